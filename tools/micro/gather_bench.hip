// gather_bench.hip — what does a divergent node fetch cost on gfx950?  (profiles/r03_notes.md, "what bounds the trace kernel")
// Every wave gathers random 32-byte "nodes" from a table (1.5 MB by default: L2-resident like the scene) in a dependent chain
// (the next index comes from the fetched data, as in a traversal), 8 waves per SIMD like the trace kernel.
//   A  lane = ray, two dwordx4 loads per step (q0, q1 of the lane's node)                     [the shipped inner step]
//   B  lane = ray, ONE dwordx4 load per step (16-byte nodes)
//   C  lane PAIR = ray: even lane loads q0, odd lane q1 of the pair's node, ONE instruction; halves exchanged by DPP
//   D  like A, but idle half of the lanes (odd) read node 0 (same address)
//   E  like A, from LDS (the first 10 KB of the table staged per block)
//   F  like A with 4 lanes of a quad on the SAME node (coherent quads)
//   G  lane = ray, a 64-byte node (4-wide tree, half-precision boxes): FOUR dwordx4 loads of one half line
//   H  lane = ray, a 64-byte node of which three quarters are read (THREE dwordx4)
//   I  lane = ray, a 128-byte node (8-wide tree, quantised boxes): SIX dwordx4 loads of one line
// Prints ns per node-visit per CU-cycle figures.  build: hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256, 8) void gather(const uint4* __restrict__ table, uint32_t nnodes, uint32_t steps, uint32_t* out)
{
    extern __shared__ uint4 s_nodes[];
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t ncache = 0;
    if (MODE == 4) {
        ncache = 320;
        for (uint32_t i = threadIdx.x; i < 2 * ncache; i += blockDim.x) s_nodes[i] = table[i];
        __syncthreads();
    }
    uint32_t ray = (MODE == 2) ? (blockIdx.x * blockDim.x + threadIdx.x) >> 1 : (MODE == 5 ? (blockIdx.x * blockDim.x + threadIdx.x) >> 2 : blockIdx.x * blockDim.x + threadIdx.x);
    uint32_t node = (ray * 2654435761u) % nnodes;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        uint4 q0, q1;
        if (MODE == 0) { q0 = table[2 * node]; q1 = table[2 * node + 1]; }
        else if (MODE == 1) { q0 = table[2 * node]; q1 = q0; }
        else if (MODE == 2) {
            const uint4 h = table[2 * node + (lane & 1u)];            // my half of the pair's node
            uint4 o;                                                   // the partner's half (quad_perm [1,0,3,2])
            o.x = __builtin_amdgcn_mov_dpp(h.x, 0xB1, 0xF, 0xF, true); o.y = __builtin_amdgcn_mov_dpp(h.y, 0xB1, 0xF, 0xF, true);
            o.z = __builtin_amdgcn_mov_dpp(h.z, 0xB1, 0xF, 0xF, true); o.w = __builtin_amdgcn_mov_dpp(h.w, 0xB1, 0xF, 0xF, true);
            q0 = (lane & 1u) ? o : h; q1 = (lane & 1u) ? h : o;
        }
        else if (MODE == 3) { const uint32_t n = (lane & 1u) ? 0u : node; q0 = table[2 * n]; q1 = table[2 * n + 1]; }
        else if (MODE == 4) { const uint32_t n = node % ncache; q0 = s_nodes[2 * n]; q1 = s_nodes[2 * n + 1]; }
        else if (MODE == 6) { const uint32_t n = node % (nnodes / 2u); q0 = table[4 * n]; q1 = table[4 * n + 1]; const uint4 q2 = table[4 * n + 2], q3 = table[4 * n + 3]; q0.w += q2.w; q1.w += q3.w; q0.x ^= q2.x; q1.y ^= q3.y; }
        else if (MODE == 7) { const uint32_t n = node % (nnodes / 2u); q0 = table[4 * n]; q1 = table[4 * n + 1]; const uint4 q2 = table[4 * n + 2]; q0.w += q2.w; q0.x ^= q2.x; }
        else if (MODE == 8) { const uint32_t n = node % (nnodes / 4u); q0 = table[8 * n]; q1 = table[8 * n + 1]; const uint4 q2 = table[8 * n + 2], q3 = table[8 * n + 3], q4 = table[8 * n + 4], q5 = table[8 * n + 5];
                              q0.w += q2.w + q4.w; q1.w += q3.w + q5.w; q0.x ^= q2.x ^ q4.x; q1.y ^= q3.y ^ q5.y; }
        else { q0 = table[2 * node]; q1 = table[2 * node + 1]; }
        acc += q0.x ^ q1.y;
        node = (q0.w + q1.w + s * 40503u + ray) % nnodes;              // dependent: the next node comes from the data
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv)
{
    const uint32_t nnodes = argc > 1 ? (uint32_t)atoi(argv[1]) : 48000u;      // x 32 B = 1.5 MB
    const uint32_t steps = 2000;
    std::vector<uint32_t> h((size_t)nnodes * 8);
    uint32_t x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 8; }
    uint4* d; uint32_t* out;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const dim3 grid(cus * 8), block(256);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[9] = { "A lane=ray, 2 x dwordx4", "B lane=ray, 1 x dwordx4", "C lane pair=ray, 1 x dwordx4 + DPP", "D as A, odd lanes read node 0", "E as A from LDS (320 nodes)", "F as A, quads share a node", "G 64-byte node, 4 x dwordx4", "H 64-byte node, 3 x dwordx4", "I 128-byte node, 6 x dwordx4" };
    for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 9; ++m) {
        CHECK(hipEventRecord(e0));
        const size_t lds = m == 4 ? 320 * 32 : 0;
        switch (m) {
            case 0: hipLaunchKernelGGL(gather<0>, grid, block, lds, 0, d, nnodes, steps, out); break;
            case 1: hipLaunchKernelGGL(gather<1>, grid, block, lds, 0, d, nnodes, steps, out); break;
            case 2: hipLaunchKernelGGL(gather<2>, grid, block, lds, 0, d, nnodes, steps, out); break;
            case 3: hipLaunchKernelGGL(gather<3>, grid, block, lds, 0, d, nnodes, steps, out); break;
            case 4: hipLaunchKernelGGL(gather<4>, grid, block, lds, 0, d, nnodes, steps, out); break;
            case 5: hipLaunchKernelGGL(gather<5>, grid, block, lds, 0, d, nnodes, steps, out); break;
            case 6: hipLaunchKernelGGL(gather<6>, grid, block, lds, 0, d, nnodes, steps, out); break;
            case 7: hipLaunchKernelGGL(gather<7>, grid, block, lds, 0, d, nnodes, steps, out); break;
            default: hipLaunchKernelGGL(gather<8>, grid, block, lds, 0, d, nnodes, steps, out); break;
        }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 0) continue;
        const double waves = (double)cus * 8 * 4;
        const double rays_per_wave = m == 2 ? 32.0 : (m == 3 ? 32.0 : (m == 5 ? 16.0 : 64.0));
        const double visits = waves * rays_per_wave * steps;          // distinct node visits
        const double wave_steps = waves * steps;
        printf("%-40s %8.3f ms   %6.2f G node visits/s   %6.1f CU-cycles per wave-step (at 2.4 GHz)\n", names[m], ms, visits / ms / 1e6, ms * 1e-3 * 2.4e9 * cus / wave_steps);
    }
    return 0;
}
