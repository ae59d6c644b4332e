// Host-side names of the shared buffer layouts, as the reference spells them
// (reference: structs.hpp:2-77, math3d.h:3-11).  They alias the C layouts of
// include/ptamd_structs.h, so a RayTracer user sees the same types and bytes.
#pragma once
#include "../../include/ptamd_structs.h"

using vec3 = pt_vec3;
using vec4 = pt_vec4;
using Bound = pt_Bound;
using TreeNode = pt_TreeNode;
using xyzn = pt_xyzn;
struct xyz { vec3 pos; };
using PrimitiveRender = pt_PrimitiveRender;
using PrimitiveRay = pt_PrimitiveRay;
using PrimitiveRayCS = pt_PrimitiveRayCS;
using Material = pt_Material;
using CB_Radix = pt_CB_Radix;
using CB_RT = pt_CB_RT;
constexpr int MORTON_CODE_LENGTH_EACH = PT_MORTON_CODE_LENGTH_EACH;
