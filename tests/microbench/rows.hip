// rows.hip -- micro-benchmark (test infrastructure): what bandwidth does an in-place
// "gather DEG random rows, touch, scatter back" pass reach on MI355X as a function of the row
// length and of the per-lane vector width?  Informs the HBM layout of the decoder.
//   hipcc --offload-arch=gfx950 -O3 rows.hip -o rows && ./rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int W> struct Vec;
template <> struct Vec<1> { using T = uint32_t; };
template <> struct Vec<2> { using T = uint2; };
template <> struct Vec<4> { using T = uint4; };

__device__ inline uint32_t mix(uint32_t a, uint32_t b) { return (a ^ (b >> 1)) + 0x01010101u; }
__device__ inline void acc(uint32_t &s, uint32_t v) { s = mix(s, v); }
__device__ inline void acc(uint32_t &s, uint2 v) { s = mix(s, v.x); s = mix(s, v.y); }
__device__ inline void acc(uint32_t &s, uint4 v) { s = mix(s, v.x); s = mix(s, v.y); s = mix(s, v.z); s = mix(s, v.w); }
__device__ inline uint32_t upd(uint32_t v, uint32_t s) { return v ^ s; }
__device__ inline uint2 upd(uint2 v, uint32_t s) { return make_uint2(v.x ^ s, v.y ^ s); }
__device__ inline uint4 upd(uint4 v, uint32_t s) { return make_uint4(v.x ^ s, v.y ^ s, v.z ^ s, v.w ^ s); }

// rows: [G][E][R bytes]; one wave handles sub-row `sub` (256*W bytes) of DEG rows of NPW nodes
template <int W, int DEG, int UNROLL>
__global__ __launch_bounds__(256) void pass(uint8_t *data, const int *idx, int E, int R, int n_nodes, int npw, int subs) {
    using T = typename Vec<W>::T;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int waves_per_group = ((n_nodes + npw - 1) / npw) * subs;
    const int g = wave / waves_per_group;
    const int r = wave - g * waves_per_group;
    const int sub = r % subs, chunk = r / subs;
    const size_t base = (size_t)g * E * R + (size_t)sub * 256 * W + (size_t)lane * 4 * W;
    int n0 = chunk * npw, n1 = min(n0 + npw, n_nodes);
    for (int n = n0; n < n1; n += UNROLL) {
        T v[UNROLL][DEG];
        int e[UNROLL][DEG];
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
#pragma unroll
            for (int k = 0; k < DEG; k++) {
                e[u][k] = __builtin_amdgcn_readfirstlane(idx[(size_t)min(n + u, n1 - 1) * DEG + k]);
                v[u][k] = *reinterpret_cast<const T *>(data + base + (size_t)e[u][k] * R);
            }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            uint32_t s = 0;
#pragma unroll
            for (int k = 0; k < DEG; k++) acc(s, v[u][k]);
            if (n + u < n1)
#pragma unroll
                for (int k = 0; k < DEG; k++) *reinterpret_cast<T *>(data + base + (size_t)e[u][k] * R) = upd(v[u][k], s);
        }
    }
}

template <int W, int DEG, int UNROLL>
void run(const char *name, uint8_t *d, const int *d_idx, int E, int R, int n_nodes, size_t total_frames, int npw) {
    const int G = (int)(total_frames / R);
    const int subs = R / (256 * W);
    const int waves = G * ((n_nodes + npw - 1) / npw) * subs;
    dim3 grid((waves + 3) / 4), block(256);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((pass<W, DEG, UNROLL>), grid, block, 0, 0, d, d_idx, E, R, n_nodes, npw, subs);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((pass<W, DEG, UNROLL>), grid, block, 0, 0, d, d_idx, E, R, n_nodes, npw, subs);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    double bytes = 2.0 * (double)n_nodes * DEG * (double)total_frames;
    printf("%-34s R=%4d W=%d U=%d npw=%3d  %8.3f ms  %7.1f GB/s\n", name, R, W, UNROLL, npw, ms, bytes / ms / 1e6);
}

int main() {
    const int DEG = 7, n_nodes = 32400, E = n_nodes * DEG;      // DVB-S2-like check pass
    const size_t frames = 4096;
    std::vector<int> perm(E);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(1);
    std::shuffle(perm.begin(), perm.end(), rng);
    std::vector<int> seq(E);
    std::iota(seq.begin(), seq.end(), 0);
    uint8_t *d;
    int *d_rand, *d_seq;
    CK(hipMalloc(&d, (size_t)E * frames));
    CK(hipMemset(d, 1, (size_t)E * frames));
    CK(hipMalloc(&d_rand, E * sizeof(int)));
    CK(hipMalloc(&d_seq, E * sizeof(int)));
    CK(hipMemcpy(d_rand, perm.data(), E * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_seq, seq.data(), E * sizeof(int), hipMemcpyHostToDevice));
    printf("in-place pass over %d nodes x %d rows, %zu frames (%.2f GB touched per pass, read+write)\n", n_nodes, DEG, frames, 2.0 * E * frames / 1e9);
    for (int npw : {8, 32}) {
        run<1, 7, 1>("random rows", d, d_rand, E, 256, n_nodes, frames, npw);
        run<1, 7, 2>("random rows", d, d_rand, E, 256, n_nodes, frames, npw);
        run<1, 7, 4>("random rows", d, d_rand, E, 256, n_nodes, frames, npw);
        run<1, 7, 2>("random rows", d, d_rand, E, 1024, n_nodes, frames, npw);
        run<2, 7, 2>("random rows", d, d_rand, E, 512, n_nodes, frames, npw);
        run<2, 7, 2>("random rows", d, d_rand, E, 1024, n_nodes, frames, npw);
        run<4, 7, 1>("random rows", d, d_rand, E, 1024, n_nodes, frames, npw);
        run<4, 7, 2>("random rows", d, d_rand, E, 1024, n_nodes, frames, npw);
        run<4, 7, 2>("random rows", d, d_rand, E, 4096, n_nodes, frames, npw);
        run<1, 7, 2>("sequential rows (VN-like)", d, d_seq, E, 256, n_nodes, frames, npw);
        run<4, 7, 2>("sequential rows (VN-like)", d, d_seq, E, 1024, n_nodes, frames, npw);
    }
    return 0;
}
