// The f32-weight half of acmi_gemm.hip as a translation unit of its own (the two halves compile in parallel: this file
// used to be the whole build's critical path).  Holds launch_rowmajor<float> / launch_tiled<float> / launch_pair<float>
// and every kernel they instantiate; nothing else.
#define ACMI_GEMM_F32_TU 1
#include "acmi_gemm.hip"
