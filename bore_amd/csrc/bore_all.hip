// bore_all.hip -- libbore_hip.so as ONE translation unit: the fused per-loop iteration kernel
// (bore_iter.hip) runs the bodies of the fit, screening and L-BFGS-B kernels back to back, so it
// has to see all of them.
#include "bore_hip.hip"
#include "bore_argmax.hip"
#include "bore_svgd.hip"
#include "bore_iter.hip"
#include "bore_engine.hip"
