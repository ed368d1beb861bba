// tcp_rates.hip -- what the gfx950 vector L1 (TCP) charges for the divergent gathers of BVH traversal.
//
// Every lane gathers records at pseudo-random positions of a buffer that fits L2 / the Infinity Cache (no HBM bound),
// 20 waves per CU like k_ao_rays.  Patterns:
//   own64    lane loads its own 64-B record with 4 x global_load_dwordx4           (the 4-wide node fetch of lv_node_step)
//   own48    3 x dwordx4 of a 48-B record                                           (candidate 48-B node)
//   own32    2 x dwordx4 of a 32-B record                                           (segment record)
//   own16    1 x dwordx4
//   own64x2  lane loads 64 B as 8 x dwordx2;  own64x1: 16 x dword                   (is the charge per request or per byte?)
//   quad64   the 4 lanes of a quad load the four 16-B chunks of ONE 64-B record     (16 distinct lines per instruction)
//   row64    16 lanes share one 64-B record's line (each reads the same 16 B chunk) (4 distinct lines per instruction)
// Reported per pattern and working-set size: lane-requests per ns, and per CU per cycle at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITERS 512

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256, 5) void k_gather(const float4* __restrict__ buf, unsigned mask16, float* out) {
    // mask16: number of 16-B chunks in the buffer - 1 (power of two)
    const unsigned tid = blockIdx.x * 256u + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    unsigned seed = tid * 9781u + 12345u;
    float acc = 0.0f;
    for (int it = 0; it < ITERS; it++) {
        seed = hash(seed + it);
        if (MODE == 0) { // own64: 4 x dwordx4
            const unsigned c = (seed & mask16) & ~3u;
            const float4 a = buf[c], b = buf[c + 1], d = buf[c + 2], e = buf[c + 3];
            acc += (a.x + b.y) + (d.z + e.w);
        } else if (MODE == 1) { // own48
            const unsigned c = ((seed & mask16) / 3u) * 3u;
            const float4 a = buf[c], b = buf[c + 1], d = buf[c + 2];
            acc += (a.x + b.y) + d.z;
        } else if (MODE == 2) { // own32
            const unsigned c = (seed & mask16) & ~1u;
            const float4 a = buf[c], b = buf[c + 1];
            acc += a.x + b.y;
        } else if (MODE == 3) { // own16
            const float4 a = buf[seed & mask16];
            acc += a.x;
        } else if (MODE == 4) { // own64 as 8 x dwordx2
            const unsigned c = (seed & mask16) & ~3u;
            const float2* q = reinterpret_cast<const float2*>(buf + c);
#pragma unroll
            for (int k = 0; k < 8; k++) { const float2 v = q[k]; acc += v.x + v.y; }
        } else if (MODE == 5) { // own64 as 16 x dword
            const unsigned c = (seed & mask16) & ~3u;
            const float* q = reinterpret_cast<const float*>(buf + c);
#pragma unroll
            for (int k = 0; k < 16; k++) acc += q[k];
        } else if (MODE == 6) { // quad64: quad shares the record, lane takes chunk lane & 3; 4 records per lane-iteration
            // (same bytes per lane as own64: 4 loads, each from a different record of the quad's four)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned qs = hash((tid >> 2) * 40503u + unsigned(it) * 4u + unsigned(k));
                const unsigned c = (qs & mask16) & ~3u;
                const float4 a = buf[c + (lane & 3u)];
                acc += a.x;
            }
        } else if (MODE == 7) { // row64: 16 lanes read the same 16-B chunk
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned qs = hash((tid >> 4) * 40503u + unsigned(it) * 4u + unsigned(k));
                const float4 a = buf[qs & mask16];
                acc += a.x;
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

typedef void (*kern_t)(const float4*, unsigned, float*);

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t maxBytes = size_t(64) << 20;
    float4* buf; float* out;
    CHECK(hipMalloc(&buf, maxBytes)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(buf, 0, maxBytes));
    struct { const char* name; kern_t fn; int reqPerIter; int bytesPerReq; } ps[] = {
        {"own64 (4 x dwordx4)", k_gather<0>, 4, 16}, {"own48 (3 x dwordx4)", k_gather<1>, 3, 16},
        {"own32 (2 x dwordx4)", k_gather<2>, 2, 16}, {"own16 (1 x dwordx4)", k_gather<3>, 1, 16},
        {"own64 (8 x dwordx2)", k_gather<4>, 8, 8}, {"own64 (16 x dword)", k_gather<5>, 16, 4},
        {"quad64 (4 x dwordx4, quad shares a 64-B record)", k_gather<6>, 4, 16},
        {"row64 (4 x dwordx4, 16 lanes share a chunk)", k_gather<7>, 4, 16},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"waves_per_cu\": 20, \"rows\": [\n", prop.gcnArchName, cus,
           prop.clockRate / 1000);
    bool first = true;
    for (size_t kb : {16, 256, 2048, 32768, 65536}) {
        const size_t mb = kb >> 10;
        const unsigned mask16 = unsigned((kb << 10) / 16 - 1);
        for (auto& p : ps) {
            const int grid = cus * 5;
            p.fn<<<grid, 256>>>(buf, mask16, out);
            CHECK(hipEventRecord(e0));
            p.fn<<<grid, 256>>>(buf, mask16, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double laneReq = double(grid) * 256.0 * ITERS * p.reqPerIter;
            const double perCuPerNs = laneReq / (double(ms) * 1e6) / cus;
            printf("%s  {\"pattern\": \"%s\", \"working_set_kb\": %zu, \"ms\": %.4f, \"lane_requests\": %.0f, "
                   "\"lane_requests_per_cu_per_ns\": %.4f, \"bytes_per_cu_per_ns\": %.2f}",
                   first ? "" : ",\n", p.name, kb, ms, laneReq, perCuPerNs, perCuPerNs * p.bytesPerReq);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
