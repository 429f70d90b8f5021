// Device code of the frame kernels, for the ahead-of-time translation units (path_tracer.hip: IEEE arithmetic, -ffp-contract=off;
// shade_fast.hip: the shading kernels a second time with the arithmetic a Vulkan implementation is allowed).  The traversal lives in
// trace.h / trace_quad.h, the shading kernels in shade_kernel.h, the state they share in pt_state.h.
#pragma once
#include "pt.h"
#include "trace.h"
#include "trace_quad.h"
#include "pt_state.h"
#include "shade_kernel.h"
#include "trace_lanes.h"
