// hostmirror_e1_capi.cpp - the host-side TraceTay of 01_Basic behind a C call for tests/test_host_mirror.py (bit for bit against
// the frame computed by the reference's own classes, tests/golden/frame_01.npz).  Test support only; no GPU.
#include "../host/01_Basic/Render.hpp"
#include "../host/01_Basic/Triangle.hpp"
#define HM_FRAME_NAME hm_e1_tracetay_frame
#include "hostmirror_whitted_frame.inc"
extern "C" void hm_e1_set_max_depth(unsigned d) { TraceTayMaxDepth() = d; }
