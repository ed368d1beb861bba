// Sensitivity probes of the node step (tools/variants.py force-includes this file for variants that set -DLV_EXP_EXTRA_LOADS=n or
// -DLV_EXP_EXTRA_VALU=n): what one more L1-hitting dwordx4 / one more VALU instruction per node step costs (EXPERIMENTS.md 3.1).
// Expanded inside lv_node_step (linevis_amd/csrc/lv_trace.h); never part of the product build.
#pragma once
#if defined(LV_EXP_EXTRA_LOADS) || defined(LV_EXP_EXTRA_VALU)
#ifndef LV_EXP_EXTRA_LOADS
#define LV_EXP_EXTRA_LOADS 0
#endif
#ifndef LV_EXP_EXTRA_VALU
#define LV_EXP_EXTRA_VALU 0
#endif
#define LV_NODE_STEP_PROBE(p, q0, inv, tMax, S)                                                                              \
    do {                                                                                                                     \
        float sink = 0.0f;                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < LV_EXP_EXTRA_LOADS; k++) {                                                     \
            const float4 x = (p)[(k & 3) + 4 * int((S).numSegs >> 31)]; /* numSegs < 2^31: same node, but not provably so */ \
            sink += (x.x + x.y) + (x.z + x.w);                                                                               \
        }                                                                                                                    \
        float e0 = (q0).x, e1 = (q0).y, e2 = (q0).z, e3 = (q0).w;                                                            \
        _Pragma("unroll") for (int k = 0; k < LV_EXP_EXTRA_VALU / 4; k++) {                                                  \
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(e0) : "v"((inv).x));                                              \
            asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(e1));                                                              \
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(e2) : "v"((inv).y));                                                  \
            asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(e3) : "v"((inv).z));                                         \
        }                                                                                                                    \
        sink += (e0 + e1) + (e2 + e3);                                                                                       \
        if (sink == 1.2345678e-30f) tMax = 0.0f; /* keeps the probes alive; never true in practice */                        \
    } while (0)
#endif
