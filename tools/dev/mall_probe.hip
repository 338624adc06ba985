// Development probe (not part of the library): can K2t's product buffer live in the 256 MiB Infinity Cache?
// K2t writes one product per entry in pass 1 and reads it back in pass 2 -- half of the 16 B (f32) an entry costs.  If the
// passes are run super-block by super-block (a range of row blocks at a time), the buffer between them is small and
// re-used; whether that keeps its bytes away from HBM is what this measures.
//   A(k): reads val (4 B) + code (2 B) of chunk k (streamed once, non-temporal), writes prod (4 B) into a buffer of `chunk` entries
//   B(k): reads that buffer (4 B) + rowc (2 B) of chunk k, folds to one value per workgroup
// Total entries N are fixed; K = number of chunks.  K = 1 is today's K2t (1.28 GB written, then read).
// Build: hipcc -O3 --offload-arch=gfx950 tools/dev/mall_probe.hip -o tools/dev/mall_probe      Run: tools/dev/mall_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

// 16 bytes of values and 8 bytes of codes per lane and step
template <int NT_ST>
__global__ __launch_bounds__(1024) void k_a(const f4 *__restrict__ val, const u2 *__restrict__ code, f4 *__restrict__ prod, uint64_t pieces) {
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < pieces; i += (uint64_t)gridDim.x * 1024) {
        const f4 v = __builtin_nontemporal_load(val + i);
        const u2 c = __builtin_nontemporal_load(code + i);
        f4 p;
        p.x = v.x * (float)(c.x & 0xFFFF); p.y = v.y * (float)(c.x >> 16);
        p.z = v.z * (float)(c.y & 0xFFFF); p.w = v.w * (float)(c.y >> 16);
        if (NT_ST) __builtin_nontemporal_store(p, prod + i); else prod[i] = p;
    }
}

template <int NT_LD>
__global__ __launch_bounds__(256) void k_b(const f4 *__restrict__ prod, const u2 *__restrict__ rowc, float *__restrict__ out, uint64_t pieces) {
    float s = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < pieces; i += (uint64_t)gridDim.x * 256) {
        const f4 p = NT_LD ? __builtin_nontemporal_load(prod + i) : prod[i];
        const u2 r = __builtin_nontemporal_load(rowc + i);
        s += p.x * (float)(r.x & 1) + p.y * (float)(r.x >> 31) + p.z * (float)(r.y & 1) + p.w * (float)(r.y >> 31);
    }
    for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}

int main(int argc, char **argv) {
    const uint64_t N = argc > 1 ? strtoull(argv[1], 0, 10) : 320000000ull;  // entries
    const uint64_t pieces = N / 4;
    f4 *val, *prod[2];
    u2 *code, *rowc;
    float *out;
    CK(hipMalloc(&val, pieces * 16));
    CK(hipMalloc(&code, pieces * 8));
    CK(hipMalloc(&rowc, pieces * 8));
    CK(hipMalloc(&prod[0], pieces * 16));
    CK(hipMalloc(&prod[1], pieces * 16));
    CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(val, 0x3c, pieces * 16));
    CK(hipMemset(code, 1, pieces * 8));
    CK(hipMemset(rowc, 1, pieces * 8));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    hipEvent_t e0, e1, ev[2];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    const int grid_a = 256 * 2, grid_b = 256 * 8;
    printf("N = %llu entries: A moves 10 B per entry (6 in, 4 out), B 6 B per entry; 16 B x N = %.2f GB\n", (unsigned long long)N, 16.0 * N / 1e9);
    printf("%-34s %6s %10s %9s %12s\n", "mode", "K", "buffer MB", "ms", "TB/s (16 B)");
    auto run = [&](const char *name, int K, int mode, int nt) {
        // mode 0: one stream, one buffer re-used by every chunk; 1: one stream, every chunk its own part of a full-size buffer
        // (the bytes go to HBM and come back, only later); 2: two streams, A(k+1) beside B(k), two buffers
        const uint64_t cp = (pieces + K - 1) / K;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s0));
            for (int k = 0; k < K; ++k) {
                const uint64_t b = (uint64_t)k * cp, n = std::min(cp, pieces - b);
                f4 *p = mode == 1 ? prod[0] + b : prod[mode == 2 ? k & 1 : 0];
                if (mode == 2) {
                    // A(k) on s0 must wait until B(k-2), the last reader of this buffer, is done
                    if (k >= 2) CK(hipStreamWaitEvent(s0, ev[k & 1], 0));
                }
                if (nt & 1) hipLaunchKernelGGL(k_a<1>, dim3(grid_a), dim3(1024), 0, s0, val + b, code + b, p, n);
                else hipLaunchKernelGGL(k_a<0>, dim3(grid_a), dim3(1024), 0, s0, val + b, code + b, p, n);
                hipStream_t sb = s0;
                if (mode == 2) {
                    hipEvent_t done;  // B(k) on s1 after A(k)
                    CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
                    CK(hipEventRecord(done, s0));
                    CK(hipStreamWaitEvent(s1, done, 0));
                    CK(hipEventDestroy(done));
                    sb = s1;
                }
                if (nt & 2) hipLaunchKernelGGL(k_b<1>, dim3(grid_b), dim3(256), 0, sb, p, rowc + b, out, n);
                else hipLaunchKernelGGL(k_b<0>, dim3(grid_b), dim3(256), 0, sb, p, rowc + b, out, n);
                if (mode == 2) CK(hipEventRecord(ev[k & 1], s1));
            }
            if (mode == 2) { CK(hipStreamWaitEvent(s0, ev[0], 0)); if (K > 1) CK(hipStreamWaitEvent(s0, ev[1], 0)); }
            CK(hipEventRecord(e1, s0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("%-34s %6d %10.1f %9.3f %12.2f\n", name, K, cp * 16.0 / 1e6, best, 16.0 * N / best / 1e9);
        fflush(stdout);
    };
    const int Ks[] = {1, 4, 8, 16, 24, 32, 48, 64, 128};
    for (int K : Ks) run("re-used buffer, one stream", K, 0, 0);
    for (int K : Ks) run("own part of a big buffer", K, 1, 0);
    for (int K : {8, 16, 32, 64}) run("re-used, nt store + nt load", K, 0, 3);
    for (int K : {8, 16, 32, 64}) run("re-used, nt load only", K, 0, 2);
    for (int K : {8, 16, 32, 64, 128}) run("two buffers, A(k+1) beside B(k)", K, 2, 0);
    return 0;
}
