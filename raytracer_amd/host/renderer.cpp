// renderer.cpp -- Renderer::Init()/Tick() over the rtmi C ABI (see renderer.h).
#include "renderer.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace rtmi {

void FillSeedTable( uint32_t* out, size_t n )
{
	uint32_t seed = 0x12345678u;
	for (size_t i = 0; i < n; i++)
	{
		seed ^= seed << 13, seed ^= seed >> 17, seed ^= seed << 5;
		out[i] = seed;
	}
}

Renderer::Renderer( int w, int h, int dev ) : width( w ), height( h ), device( dev ) { camera = new Camera( w, h ); }

Renderer::~Renderer()
{
	if (ctx) rtmi_destroy( ctx );
	free( accumulator ), free( screenPixels );
	delete camera;
}

bool Renderer::Init()
{
	const size_t n = (size_t)width * height;
	accumulator = (float4*)aligned_alloc( 64, n * 16 ); // renderer.cpp:10-11
	memset( (void*)accumulator, 0, n * 16 );
	screenPixels = (uint32_t*)aligned_alloc( 64, ( n * 4 + 63 ) & ~(size_t)63 );
	memset( screenPixels, 0, n * 4 );
	int r = rtmi_create( &ctx, device, width, height );
	if (r != RTMI_OK) { error = "rtmi_create failed (" + std::to_string( r ) + "): no usable HIP device / library"; ctx = 0; return false; }
	seeds.resize( n );
	FillSeedTable( seeds.data(), n ); // renderer.cpp:51-57
	if (rtmi_set_seeds( ctx, seeds.data() ) != RTMI_OK || rtmi_set_camera( ctx, &camera->gpuCamProp[0].x ) != RTMI_OK) { error = rtmi_last_error( ctx ); return false; }
	if (scene.isInitialized && !UploadScene()) return false;
	return true;
}

bool Renderer::UploadScene()
{
	if (!ctx) { error = "UploadScene before Init"; return false; }
	if (rtmi_upload_scene( ctx, scene.Desc() ) != RTMI_OK) { error = rtmi_last_error( ctx ); return false; }
	scene.lightsChanged = false; // (the upload carries the lights)
	ResetAccumulation();
	return true;
}

void Renderer::ResetAccumulation() // renderer.cpp:622-633
{
	accumulatedFrames = 0;
	if (accumulator) memset( (void*)accumulator, 0, (size_t)width * height * 16 );
	if (ctx) rtmi_reset_accumulator( ctx ), rtmi_set_camera( ctx, &camera->gpuCamProp[0].x );
}

void Renderer::SetVisualizationMode( VisualizationMode visual )
{
	visualizationMode = visual;
	if (visual == PathTracing) ResetAccumulation();
}

void Renderer::SetGPU( bool mode )
{
	if (!mode) { error = "SetGPU(false): this build has no CPU trace path"; return; }
	useGPU = true;
	ResetAccumulation();
}

bool Renderer::ReadAccumulator()
{
	if (!ctx || rtmi_read_accumulator( ctx, &accumulator[0].x ) != RTMI_OK) { if (ctx) error = rtmi_last_error( ctx ); return false; }
	return true;
}

void Renderer::Tick( float )
{
	if (!scene.isInitialized || !ctx) return; // renderer.cpp:448-449
	auto t0 = std::chrono::high_resolution_clock::now();
	int r;
	if (scene.lightsChanged) // Scene::SetLight* since the last frame (scene.h:536-551)
	{
		const rtmi_scene_desc* d = scene.Desc();
		if (rtmi_set_whitted_lights( ctx, &d->whitted_light_pos[0][0], &d->whitted_light_color[0][0], d->whitted_intensity ) != RTMI_OK) error = rtmi_last_error( ctx );
		scene.lightsChanged = false;
	}
	if (visualizationMode == PathTracing)
	{
		// renderer.cpp:524-556: one more frame of samples, then Finalize with f = accumulatedFrames
		r = rtmi_render( ctx, RTMI_PATHTRACING, (int)accumulatedFrames, samplesPerTick, 1, recursionDepth, 0 );
		accumulatedFrames += samplesPerTick;
		if (r == RTMI_OK) r = rtmi_finalize( ctx, screenPixels, (int)accumulatedFrames, postfx_on ? &postfx : 0 );
	}
	else
	{
		// renderer.cpp:578-618: accumulator[p] = Trace(primary ray, 0); RGBF32_to_RGB8 without frame count
		r = rtmi_render( ctx, (int)visualizationMode, 0, 1, 1, recursionDepth, anti_aliasing ? 1 : 0 );
		if (r == RTMI_OK) r = rtmi_finalize( ctx, screenPixels, 0, postfx_on ? &postfx : 0 );
	}
	if (r != RTMI_OK) error = rtmi_last_error( ctx );
	if (camera->Update() && visualizationMode == PathTracing) ResetAccumulation(); // renderer.cpp:622-633
	else if (ctx) rtmi_set_camera( ctx, &camera->gpuCamProp[0].x );
	// renderer.cpp:635-641
	lastFrameMs = std::chrono::duration<float, std::milli>( std::chrono::high_resolution_clock::now() - t0 ).count();
	avgMs = ( 1 - alpha ) * avgMs + alpha * lastFrameMs;
	if (alpha > 0.05f) alpha *= 0.5f;
}

// GLFW key codes as the reference uses them (renderer.cpp:368-441)
enum { KEY_A = 65, KEY_D = 68, KEY_E = 69, KEY_Q = 81, KEY_S = 83, KEY_W = 87, KEY_RIGHT = 262, KEY_LEFT = 263, KEY_DOWN = 264, KEY_UP = 265 };
void Renderer::KeyDown( int key )
{
	const float velocity = .05f;
	switch (key)
	{
	case KEY_A: camera->MoveHorizontal( -velocity, Camera::Left ); break;
	case KEY_D: camera->MoveHorizontal( velocity, Camera::Right ); break;
	case KEY_W: camera->MoveVertical( velocity, Camera::Up ); break;
	case KEY_S: camera->MoveVertical( -velocity, Camera::Down ); break;
	case KEY_E: camera->MoveDistal( velocity, Camera::In ); break;
	case KEY_Q: camera->MoveDistal( -velocity, Camera::Out ); break;
	case KEY_UP: camera->rotateDirections[2] = true; break;
	case KEY_DOWN: camera->rotateDirections[3] = true; break;
	case KEY_LEFT: camera->rotateDirections[0] = true; break;
	case KEY_RIGHT: camera->rotateDirections[1] = true; break;
	default: break;
	}
}
void Renderer::KeyUp( int key )
{
	switch (key)
	{
	case KEY_A: camera->Translate( Camera::Left ); break;
	case KEY_D: camera->Translate( Camera::Right ); break;
	case KEY_W: camera->Translate( Camera::Up ); break;
	case KEY_S: camera->Translate( Camera::Down ); break;
	case KEY_E: camera->Translate( Camera::In ); break;
	case KEY_Q: camera->Translate( Camera::Out ); break;
	case KEY_UP: camera->rotateDirections[2] = false; break;
	case KEY_DOWN: camera->rotateDirections[3] = false; break;
	case KEY_LEFT: camera->rotateDirections[0] = false; break;
	case KEY_RIGHT: camera->rotateDirections[1] = false; break;
	default: break;
	}
}

} // namespace rtmi
