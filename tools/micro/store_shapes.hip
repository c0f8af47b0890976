// How fast does the chip take a 1 GB row-major f32 matrix [M][256] written in the shapes a GEMM epilogue can offer?
// One 16-byte store per lane; a wave instruction covers SEG bytes of each of 1024 / SEG rows (row pitch 1 KB).
//   SEG = 128: 8 rows x 128 B (a 32 x 32 accumulator block turned through LDS: lkg_gemm_tall's plain epilogue)
//   SEG = 256: 4 rows x 256 B      SEG = 512: 2 rows x 512 B      SEG = 1024: one whole row
// Workgroups of 512 threads own 128 x 256 tiles (like the GEMM) and write them in the order the epilogue does.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_shapes.hip -o /tmp/store_shapes && /tmp/store_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float nt4 __attribute__((ext_vector_type(4)));

template <int SEG, bool NT>
__global__ __launch_bounds__(512) void store_kernel(float *c, long m, int tiles_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = SEG / 16;            // lanes per row segment
    constexpr int RPI = 64 / LPR;            // rows per instruction
    constexpr int CG = 1024 / SEG;           // column groups of a row
    // a wave owns rows [wm * 64, +64) x column group(s) like the GEMM's 2 x 4 wave grid when SEG <= 256; for wider segments
    // the waves split the tile's 128 rows evenly
    for (int t = 0; t < tiles_per_wg; ++t) {
        const long tile = (long)blockIdx.x * tiles_per_wg + t;
        const long m0 = tile * 128;
        if (m0 >= m) return;
        // 128 rows x CG groups = 128 * CG (row, group) units; RPI rows per instruction; 8 waves
        const int instr_total = 128 * CG / RPI;          // per tile
        for (int k = wave; k < instr_total; k += 8) {
            // instruction k: column group k % CG ... keep consecutive instructions of a wave in the same column group
            const int cg = (k / 8) % CG;
            const int rb = ((k / 8) / CG) * 8 + (k % 8);
            const long row = m0 + (long)rb * RPI + lane / LPR;
            const int col = cg * (SEG / 4) + 4 * (lane % LPR);
            if (row < m) {
                nt4 v = {(float)row, (float)col, 1.f, 2.f};
                nt4 *dst = reinterpret_cast<nt4 *>(c + row * 256 + col);
                if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        }
    }
}

template <int SEG, bool NT>
float run(float *c, long m, int grid = 0) {
    const int tiles = (int)((m + 127) / 128);
    const int tpw = grid ? (tiles + grid - 1) / grid : 16;
    if (!grid) grid = (tiles + tpw - 1) / tpw;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ts;
    for (int it = 0; it < 7; ++it) {
        hipEventRecord(a);
        store_kernel<SEG, NT><<<grid, 512>>>(c, m, tpw);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main() {
    const long m = 1000000;
    float *c;
    hipMalloc(&c, m * 256 * 4);
    const double gb = m * 256 * 4 / 1e9;
#define GO(SEG, NT) { float ms = run<SEG, NT>(c, m); printf("segment %4d B  %s  %.3f ms  %.2f TB/s\n", SEG, NT ? "nt   " : "plain", ms, gb / ms); }
    GO(128, true) GO(256, true) GO(512, true) GO(1024, true)
    GO(128, false) GO(256, false) GO(512, false) GO(1024, false)
    // a per-CU limit?  the same 1 GB written by FEWER workgroups (one per CU up to 256, then two): GB/s per workgroup
    for (int grid : {16, 32, 64, 128, 256, 512}) {
        float ms = run<128, true>(c, m, grid);
        printf("%4d workgroups of 512 threads: %.3f ms  %.2f TB/s  %.1f GB/s per workgroup\n", grid, ms, gb / ms, gb / ms * 1e3 / grid);
    }
    hipFree(c);
    return 0;
}
