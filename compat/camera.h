// compat/camera.h -- camera.h:11-168 of the reference: rtmi::Camera at the compile-time resolution the reference uses
// (SCRWIDTH x SCRHEIGHT, camera.h:4-5).  gpuCamProp is what rtmi_set_camera takes (camera.h:25, renderer.cpp:63).
#pragma once
#include "precomp.h"
namespace Tmpl8 {
class Camera : public rtmi::Camera
{
public:
	Camera() : rtmi::Camera( SCRWIDTH, SCRHEIGHT ) {}
};
}
