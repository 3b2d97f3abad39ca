// compat/renderer.h -- renderer.h:1-125 of the reference with the device members replaced: one rtmi_ctx* instead of the 6 Kernel*
// and ~25 Buffer* (renderer.h:80-123).  Init() and Tick() are what INTEGRATION.md 1-3 write out; a TU that holds them is
// tests/compat/integration_tu.cpp.  TheApp is the template's application base (template/common.h / precomp.h: `screen`, the flow
// and input methods).
#pragma once
#include "camera.h"
#include "scene.h"
namespace Tmpl8 {

class TheApp
{
public:
	virtual ~TheApp() = default;
	virtual void Init() = 0;
	virtual void Tick( float deltaTime ) = 0;
	virtual void Shutdown() = 0;
	virtual void MouseUp( int button ) = 0;
	virtual void MouseDown( int button ) = 0;
	virtual void MouseMove( int x, int y ) = 0;
	virtual void MouseWheel( float y ) = 0;
	virtual void KeyUp( int key ) = 0;
	virtual void KeyDown( int key ) = 0;
	Surface* screen = 0;
};

struct int2 { int x = 0, y = 0; };

class Renderer : public TheApp
{
public:
	enum VisualizationMode { Normal, Distance, Albedo, RayTracing, PathTracing }; // == rtmi_mode
	void Init();
	void Tick( float deltaTime );
	void Shutdown() {}
	void MouseUp( int ) {}
	void MouseDown( int ) {}
	void MouseMove( int x, int y ) { mousePos.x = x, mousePos.y = y; }
	void MouseWheel( float ) {}
	void KeyUp( int ) {}
	void KeyDown( int ) {}
	int2 mousePos;
	float4* accumulator = 0;
	Scene scene;
	Camera* camera = new Camera();
	bool anti_aliasing = false;
	VisualizationMode visualizationMode = RayTracing;
	int recursionDepth = 5;
	uint accumulatedFrames = 0;
	void SetAntiAliasing( bool AA ) { anti_aliasing = AA; }
	void SetVisualizationMode( VisualizationMode visual ) // renderer.h:43-58: the memset + CopyToDevice pair is one call
	{
		visualizationMode = visual;
		if (visual == VisualizationMode::PathTracing) accumulatedFrames = 0, rtmi_reset_accumulator( ctx ), rtmi_set_camera( ctx, &camera->gpuCamProp[0].x );
	}
	void SetRecusionDepth( int newDepth ) { recursionDepth = newDepth; }
	bool useGPU = true;
	void SetGPU( bool mode ) // renderer.h:64-77
	{
		useGPU = mode;
		accumulatedFrames = 0, rtmi_reset_accumulator( ctx );
		if (mode) rtmi_set_camera( ctx, &camera->gpuCamProp[0].x );
	}
	rtmi_ctx* ctx = nullptr;
	uint* seeds = 0;
	bool sceneUploaded = false;
};

} // namespace Tmpl8
