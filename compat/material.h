// compat/material.h -- material.h:3-24 of the reference: the class is rtmi::Material (same enum, fields, constructor)
#pragma once
#include "precomp.h"
namespace Tmpl8 {
using rtmi::Material;
}
