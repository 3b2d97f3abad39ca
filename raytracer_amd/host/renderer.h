// renderer.h -- the drop-in `Renderer` surface (counterpart of the reference's renderer.h / renderer.cpp).
//
// Same public face as the reference class: Init(), Tick(deltaTime), the setters the UI drives, `scene`, `camera`,
// `accumulator`, `visualizationMode`, `recursionDepth`, `accumulatedFrames`, `useGPU`.  Where the reference's
// Init()/Tick() create OpenCL Kernel/Buffer objects and enqueue kernels (renderer.cpp:14-68, 453-557), this one
// calls the C ABI of include/rtmi.h.  There is no CPU fallback in this class: if the HIP library cannot create a
// context Init() fails and Tick() does nothing but report the error.
#pragma once
#include "scene.h"
#include <string>

namespace rtmi {

class Renderer
{
public:
	enum VisualizationMode { Normal, Distance, Albedo, RayTracing, PathTracing }; // renderer.h:9
	Renderer( int width = 1280, int height = 720, int device = 0 );
	~Renderer();
	bool Init();                    // renderer.cpp:6-70: context, accumulator, seed table, camera + scene upload
	void Tick( float deltaTime );   // renderer.cpp:446-643: one frame into screenPixels
	void Shutdown() {}
	bool UploadScene();             // call after changing `scene` (first-tick upload, renderer.cpp:453-513)
	void SetAntiAliasing( bool AA ) { anti_aliasing = AA; }
	void SetVisualizationMode( VisualizationMode visual );  // renderer.h:43-58
	void SetRecusionDepth( int newDepth ) { recursionDepth = newDepth; }
	void SetGPU( bool mode );       // renderer.h:64-77; the CPU path does not exist here, `false` is refused
	void SetPostFx( const rtmi_postfx& fx, bool enabled ) { postfx = fx, postfx_on = enabled; }
	void KeyDown( int key );
	void KeyUp( int key );
	const char* LastError() const { return error.c_str(); }

	int width, height, device;
	float4* accumulator = 0;        // host mirror, refreshed by ReadAccumulator()
	bool ReadAccumulator();
	uint32_t* screenPixels = 0;     // screen->pixels
	Scene scene;
	Camera* camera = 0;
	bool anti_aliasing = false;
	VisualizationMode visualizationMode = RayTracing; // renderer.h:38
	int recursionDepth = 5;                           // renderer.h:39
	uint accumulatedFrames = 0;                       // renderer.h:40
	bool useGPU = true;                               // renderer.h:62
	int samplesPerTick = 1;         // path tracing samples per Tick (the reference renders 1 per frame)
	float lastFrameMs = 0, avgMs = 10, alpha = 1;     // renderer.cpp:635-641
	rtmi_ctx* ctx = 0;
	std::vector<uint32_t> seeds;

private:
	void ResetAccumulation();
	rtmi_postfx postfx = {};
	bool postfx_on = false;
	std::string error;
};

// first n outputs of the template RNG from its initial state 0x12345678 (template.cpp:870, 883-889; renderer.cpp:51-55)
void FillSeedTable( uint32_t* out, size_t n );

} // namespace rtmi
