// frames.h -- frames in flight on several handles of one GPU (header only; no counterpart in the reference, which has one
// device and renders one sample per Tick).
//
// A render that delivers FRAMES -- N samples per pixel split over the ranks, reduced, shown, then the next (BASELINE.json
// configs[3]) -- ends every frame with a handful of paths that have thousands of bounces left: a serial chain of ~9 us per
// bounce during which the GPU is all but idle.  FramePipeline gives consecutive frames to different rtmi_ctx
// handles, each driven by a host thread of its own, so that one frame's end runs beside the bulk of the next ones.  A frame is
// rendered, finished and delivered entirely by ONE handle with exactly the calls an unpipelined caller makes, so every delivered
// frame has the unpipelined frame's bits.  The frames' collectives (one reduce each) must be issued in the same order on every
// rank: a job takes `FramePipeline::Ordered turn( pipe, frame )` around its reduce.
#pragma once
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace rtmi {

class FramePipeline
{
public:
	// job( handle, frame ): handle = frame % handles; returns 0 or an error status
	using Job = std::function<int( int, uint64_t )>;
	explicit FramePipeline( int handles ) : lanes( handles < 1 ? 1 : handles )
	{
		for (size_t k = 0; k < lanes.size(); k++) lanes[k].th = std::thread( [this, k] { Work( (int)k ); } );
	}
	~FramePipeline()
	{
		Finish();
		{ std::lock_guard<std::mutex> lock( m ); closing = true; }
		cv.notify_all();
		for (auto& l : lanes) l.th.join();
	}
	FramePipeline( const FramePipeline& ) = delete;
	FramePipeline& operator=( const FramePipeline& ) = delete;
	// frames are numbered in submission order; a handle's frames run one after the other
	uint64_t Submit( Job job )
	{
		std::lock_guard<std::mutex> lock( m );
		const uint64_t frame = submitted++;
		lanes[frame % lanes.size()].todo.push_back( { frame, std::move( job ) } );
		cv.notify_all();
		return frame;
	}
	// waits for every submitted frame; the first non-zero job status (0: all delivered)
	int Finish()
	{
		std::unique_lock<std::mutex> lock( m );
		cv.wait( lock, [this] { return done == submitted; } );
		return status;
	}
	// the collective section of a frame: entered in frame order, left at the end of the scope
	class Ordered
	{
	public:
		Ordered( FramePipeline& p, uint64_t frame ) : p( p ), frame( frame )
		{
			std::unique_lock<std::mutex> lock( p.m );
			p.cv.wait( lock, [&] { return p.turn == frame; } );
			p.lanes[frame % p.lanes.size()].took = true;
		}
		~Ordered()
		{
			{ std::lock_guard<std::mutex> lock( p.m ); p.turn = frame + 1; }
			p.cv.notify_all();
		}
	private:
		FramePipeline& p;
		uint64_t frame;
	};

private:
	struct Item { uint64_t frame; Job job; };
	struct Lane { std::thread th; std::deque<Item> todo; bool took = false; };
	void Work( int k )
	{
		for (;;)
		{
			Item it;
			{
				std::unique_lock<std::mutex> lock( m );
				cv.wait( lock, [&] { return closing || !lanes[k].todo.empty(); } );
				if (lanes[k].todo.empty()) return;
				it = std::move( lanes[k].todo.front() ), lanes[k].todo.pop_front();
				lanes[k].took = false;
			}
			const int r = it.job( k, it.frame );
			bool took;
			{ std::lock_guard<std::mutex> lock( m ); took = lanes[k].took; }
			if (!took) { Ordered pass( *this, it.frame ); } // a job without a collective, or one that failed before it: the turn still passes on
			{
				std::lock_guard<std::mutex> lock( m );
				if (r && !status) status = r;
				done++;
			}
			cv.notify_all();
		}
	}
	std::mutex m;
	std::condition_variable cv;
	std::vector<Lane> lanes;
	uint64_t submitted = 0, done = 0, turn = 0;
	int status = 0;
	bool closing = false;
};

} // namespace rtmi
