// valu_rates.hip -- issue-rate calibration of the gfx950 VALU for the instruction classes the traversal kernels use.
//
// Each kernel runs NITER iterations of UNROLL back-to-back instructions of one class on 8 independent register chains
// per lane, at a chosen number of waves per SIMD.  Reported: wave-instructions per cycle per SIMD (s_memtime cycles of
// the slowest wave and wall-clock), i.e. the ceiling "SQ_INSTS_VALU / SIMD / cycle" can reach for that class.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o gpurun_out/valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define NITER 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 8 independent chains v[k]; every asm statement reads and writes its own chain only
#define BODY_BEGIN                                                                          \
    float v[8]; float a = p[threadIdx.x & 7], b = p[(threadIdx.x + 1) & 7];                 \
    const unsigned long long msk = __ballot(p[threadIdx.x & 63] > 1.03f);  /* wave-uniform mask in SGPRs */ \
    for (int k = 0; k < 8; k++) v[k] = p[(threadIdx.x + k) & 63] + a;  /* loads retire here */ \
    unsigned long long t0 = __builtin_readcyclecounter();                                   \
    for (int it = 0; it < NITER; it++) {
#define BODY_END                                                                            \
    }                                                                                       \
    unsigned long long t1 = __builtin_readcyclecounter();                                   \
    float s = 0; for (int k = 0; k < 8; k++) s += v[k];                                     \
    if (s == 123.456f) out[0] = s;                                                          \
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);

#define KERNEL(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void NAME(const float* p, float* out, unsigned long long* cyc) { \
        BODY_BEGIN                                                                                     \
        _Pragma("unroll") for (int u = 0; u < 4; u++) { REP8(ASM) }                                    \
        BODY_END                                                                                       \
    }

#define A_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define A_MUL(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
#define A_ADD(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
#define A_MAX(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
#define A_MAX3(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define A_MIN3(k) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define A_CVTUB(k) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(v[k]));
#define A_CVTU(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[k]));
#define A_CNDMASK(k) asm volatile("s_mov_b64 vcc, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a), "s"(msk) : "vcc");
#define A_CNDMASKS(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "s"(msk));
#define A_FMA_MAX3(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_max3_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define A_FMA_CMP(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_cmp_lt_f32 s[20:21], %0, %1" : "+v"(v[k]) : "v"(a), "v"(b) : "s20", "s21");
#define A_FMA_CND(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(v[k]) : "v"(a), "v"(b), "s"(msk));
#define A_FMA_MUL(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_mul_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a), "v"(b));
#define A_FMA_AND(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_and_b32 %0, %0, %1" : "+v"(v[k]) : "v"(a), "v"(b));
#define A_CVT_MAX(k) asm volatile("v_cvt_f32_ubyte1 %0, %0\n v_max_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
#define A_MAX_CMP(k) asm volatile("v_max_f32 %0, %0, %1\n v_cmp_lt_f32 s[20:21], %0, %1" : "+v"(v[k]) : "v"(a) : "s20", "s21");
#define A_FMA_SALU(k) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_add_u32 s20, s20, 1" : "+v"(v[k]) : "v"(a), "v"(b) : "s20", "scc");
#define A_NODEMIX(k) asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_fma_f32 %0, %0, %1, %2\n v_cvt_f32_ubyte2 %0, %0\n v_fma_f32 %0, %0, %1, %2\n " \
                                  "v_max3_f32 %0, %0, %1, %2\n v_min3_f32 %0, %0, %1, %2\n v_cmp_le_f32 s[20:21], %0, %1\n "             \
                                  "v_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(v[k]) : "v"(a), "v"(b), "s"(msk) : "s20", "s21");
#define A_FMAC(k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define A_CMP(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[k]), "v"(a) : "vcc");
#define A_CMPS(k) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(v[k]), "v"(a) : "s20", "s21");
#define A_AND(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
#define A_LSHR(k) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(v[k]));
#define A_BFE(k) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(v[k]));
#define A_ADDU(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
#define A_MOV(k) asm volatile("v_mov_b32 %0, %1" : "+v"(v[k]) : "v"(a));
#define A_MOVDPP(k) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[k]));
#define A_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[k]));
#define A_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[k]));
#define A_RSQ(k) asm volatile("v_rsq_f32 %0, %0" : "+v"(v[k]));
#define A_MULLO(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
#define A_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[k]) : "v"(a));
#define A_XOR3(k) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define A_DIVSCALE(k) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(v[k]) : "v"(a) : "vcc");
#define A_DIVFIXUP(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define A_DIVFMAS(k) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b) : );
#define A_PERM(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
#define A_READLANE(k) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(v[k]) : "s20");
#define A_BPERM(k) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(v[k]) : "v"(a));
#define A_SALU(k) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
#define A_MIX(k) asm volatile("v_cvt_f32_ubyte1 %0, %0\n v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));

KERNEL(k_fma, A_FMA) KERNEL(k_mul, A_MUL) KERNEL(k_add, A_ADD) KERNEL(k_max, A_MAX) KERNEL(k_max3, A_MAX3)
KERNEL(k_min3, A_MIN3) KERNEL(k_cvt_ubyte, A_CVTUB) KERNEL(k_cvt_u32, A_CVTU) KERNEL(k_cndmask, A_CNDMASK)
KERNEL(k_cmp_vcc, A_CMP) KERNEL(k_cmp_sgpr, A_CMPS) KERNEL(k_and, A_AND) KERNEL(k_lshr, A_LSHR) KERNEL(k_bfe, A_BFE)
KERNEL(k_add_u32, A_ADDU) KERNEL(k_mov, A_MOV) KERNEL(k_mov_dpp, A_MOVDPP) KERNEL(k_rcp, A_RCP) KERNEL(k_sqrt, A_SQRT)
KERNEL(k_rsq, A_RSQ) KERNEL(k_mul_lo_u32, A_MULLO) KERNEL(k_lshl_add, A_LSHLADD) KERNEL(k_xor3, A_XOR3)
KERNEL(k_div_scale, A_DIVSCALE) KERNEL(k_div_fixup, A_DIVFIXUP) KERNEL(k_div_fmas, A_DIVFMAS) KERNEL(k_perm, A_PERM)
KERNEL(k_cndmask_sgpr, A_CNDMASKS) KERNEL(k_fma_max3, A_FMA_MAX3) KERNEL(k_fma_cmp, A_FMA_CMP) KERNEL(k_fma_cnd, A_FMA_CND)
KERNEL(k_fma_mul, A_FMA_MUL) KERNEL(k_fma_and, A_FMA_AND) KERNEL(k_cvt_max, A_CVT_MAX) KERNEL(k_max_cmp, A_MAX_CMP)
KERNEL(k_fma_salu, A_FMA_SALU) KERNEL(k_nodemix, A_NODEMIX) KERNEL(k_fmac, A_FMAC)
KERNEL(k_readlane, A_READLANE) KERNEL(k_bpermute, A_BPERM) KERNEL(k_salu, A_SALU) KERNEL(k_cvt_fma_pair, A_MIX)

// packed fp32: two chains per instruction (register pairs)
__global__ __launch_bounds__(256) void k_pk_fma(const float* p, float* out, unsigned long long* cyc) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v[8]; f2 a = {p[threadIdx.x & 7], p[3]}, b = {p[(threadIdx.x + 1) & 7], p[5]};
    for (int k = 0; k < 8; k++) v[k] = f2{p[(threadIdx.x + k) & 63] + a.x, p[k] + a.y};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < NITER; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int k = 0; k < 8; k++) s += v[k].x + v[k].y;
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);
}
__global__ __launch_bounds__(256) void k_pk_mul(const float* p, float* out, unsigned long long* cyc) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v[8]; f2 a = {p[threadIdx.x & 7], p[3]};
    for (int k = 0; k < 8; k++) v[k] = f2{p[(threadIdx.x + k) & 63] + a.x, p[k] + a.y};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < NITER; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int k = 0; k < 8; k++) s += v[k].x + v[k].y;
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) atomicMax(cyc, t1 - t0);
}

typedef void (*kern_t)(const float*, float*, unsigned long long*);
struct Entry { const char* name; kern_t fn; int perStmt; };

int main(int argc, char** argv) {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    float* p; float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&p, 64 * 4)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&cyc, 8));
    std::vector<float> hp(64);
    for (int i = 0; i < 64; i++) hp[i] = 1.0f + 0.001f * float(i);
    CHECK(hipMemcpy(p, hp.data(), 256, hipMemcpyHostToDevice));
    Entry es[] = {
        {"v_fma_f32", k_fma, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1}, {"v_max_f32", k_max, 1},
        {"v_max3_f32", k_max3, 1}, {"v_min3_f32", k_min3, 1}, {"v_cvt_f32_ubyte1", k_cvt_ubyte, 1},
        {"v_cvt_f32_u32", k_cvt_u32, 1}, {"s_mov vcc + v_cndmask_b32(vcc)", k_cndmask, 1}, {"v_cmp_lt_f32(vcc)", k_cmp_vcc, 1},
        {"v_cmp_lt_f32(sgpr)", k_cmp_sgpr, 1}, {"v_and_b32", k_and, 1}, {"v_lshrrev_b32", k_lshr, 1}, {"v_bfe_u32", k_bfe, 1},
        {"v_add_u32", k_add_u32, 1}, {"v_mov_b32", k_mov, 1}, {"v_mov_b32_dpp(quad_perm)", k_mov_dpp, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_rsq_f32", k_rsq, 1}, {"v_mul_lo_u32", k_mul_lo_u32, 1},
        {"v_lshl_add_u32", k_lshl_add, 1}, {"v_xad_u32", k_xor3, 1}, {"v_div_scale_f32", k_div_scale, 1},
        {"v_div_fixup_f32", k_div_fixup, 1}, {"v_div_fmas_f32", k_div_fmas, 1}, {"v_perm_b32", k_perm, 1},
        {"v_readlane_b32", k_readlane, 1}, {"ds_bpermute_b32+wait", k_bpermute, 1}, {"s_add_u32", k_salu, 1},
        {"v_cvt_f32_ubyte1+v_fma_f32", k_cvt_fma_pair, 2}, {"v_cndmask_b32_e64(sgpr mask)", k_cndmask_sgpr, 1},
        {"v_fma_f32+v_max3_f32", k_fma_max3, 2}, {"v_fma_f32+v_cmp(sgpr)", k_fma_cmp, 2}, {"v_fma_f32+v_cndmask_e64", k_fma_cnd, 2},
        {"v_fma_f32+v_mul_f32", k_fma_mul, 2}, {"v_fma_f32+v_and_b32", k_fma_and, 2}, {"v_cvt_f32_ubyte1+v_max_f32", k_cvt_max, 2},
        {"v_max_f32+v_cmp(sgpr)", k_max_cmp, 2}, {"v_fma_f32+s_add_u32", k_fma_salu, 2},
        {"node-step mix: cvt,fma,cvt,fma,max3,min3,cmp,cndmask", k_nodemix, 8}, {"v_fmac_f32", k_fmac, 1}, {"v_pk_fma_f32", k_pk_fma, 1}, {"v_pk_mul_f32", k_pk_mul, 1},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"rows\": [\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    bool first = true;
    for (const Entry& e : es) {
        for (int wps : {1, 2, 4, 5, 8}) { // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
            const int grid = cus * wps;
            e.fn<<<grid, 256>>>(p, out, cyc); // warm-up
            CHECK(hipMemset(cyc, 0, 8));
            CHECK(hipEventRecord(e0));
            e.fn<<<grid, 256>>>(p, out, cyc);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long c; CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            const double insts = double(NITER) * 32.0 * e.perStmt; // per wave
            // readcyclecounter = s_memtime (100 MHz-ish constant clock on some parts) -> also report wall based
            const double perSimd = insts * wps;
            printf("%s  {\"inst\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"memtime_ticks\": %llu, "
                   "\"wave_insts_per_simd\": %.0f, \"ns_per_inst_per_simd\": %.4f}",
                   first ? "" : ",\n", e.name, wps, ms, c, perSimd, double(ms) * 1e6 / perSimd);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
