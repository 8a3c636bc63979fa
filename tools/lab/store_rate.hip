// lab: how fast can ONE CU push a 256 x 256 bf16 output tile (128 KiB) to memory, by store shape, CUs active and waves per CU?
// (round 4: the streaming GEMM's epilogue sits in its output stores for ~20 k cycles of a 64 k-cycle tile at K = 768)
//   shape 0: the GEMM epilogue's: one instruction = 8 rows x 128 B (16 B per lane), rows ld bytes apart
//   shape 1: one instruction = 2 rows x 512 B (a tile row's 256 columns are contiguous)
//   shape 2: one instruction = 1 KiB contiguous (the tile as one 128 KiB block): upper bound
//   shape 3: shape 0 with 4 rows x 256 B
// build: hipcc -O3 --offload-arch=gfx950 tools/lab/store_rate.hip -o tools/lab/store_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int SHAPE, int NT>
__global__ __launch_bounds__(1024) void store_kernel(char* out, int64_t ld, int ntn, int tiles, int nwaves, int nap) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  const int per_wave = 128 / nwaves;                  // store instructions per wave and tile
  for (int t = 0; t < tiles; ++t) {
    const int64_t tile = (int64_t)blockIdx.x + (int64_t)t * gridDim.x;
    const int64_t tm = tile / ntn, tn = tile % ntn;
    char* base = out + tm * 256 * ld + tn * 512;
    for (int i = 0; i < per_wave; ++i) {
      const int q = wave * per_wave + i;               // 0..127: which KiB of the tile
      char* p;
      if (SHAPE == 0) {
        const int rb = q >> 2, cb = q & 3;             // 32 row blocks of 8 rows x 4 column blocks of 128 B
        p = base + (int64_t)(rb * 8 + (lane >> 3)) * ld + cb * 128 + (lane & 7) * 16;
      } else if (SHAPE == 1) {
        p = base + (int64_t)(q * 2 + (lane >> 5)) * ld + (lane & 31) * 16;
      } else if (SHAPE == 3) {
        const int rb = q >> 1, cb = q & 1;
        p = base + (int64_t)(rb * 4 + (lane >> 4)) * ld + cb * 256 + (lane & 15) * 16;
      } else {
        p = out + tile * 131072 + q * 1024 + lane * 16;
      }
      if (NT) __builtin_nontemporal_store(v, (u32x4*)p);
      else *(u32x4*)p = v;
      if (nap && (i & 1)) for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(1);
    }
  }
}
template <int SHAPE, int NT>
static void run(const char* name, char* out, int64_t M, int64_t N, int grid, int nwaves, int nap) {
  const int ntn = (int)(N / 256);
  const int64_t ntiles = (M / 256) * ntn;
  const int tiles = (int)(ntiles / grid);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((store_kernel<SHAPE, NT>), dim3(grid), dim3(nwaves * 64), 0, 0, out, N * 2, ntn, tiles, nwaves, nap);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((store_kernel<SHAPE, NT>), dim3(grid), dim3(nwaves * 64), 0, 0, out, N * 2, ntn, tiles, nwaves, nap);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const double bytes = (double)tiles * grid * 131072.0;
  printf("%-34s N=%-5lld CUs=%-3d waves=%-2d nap=%-2d: %7.3f ms  %6.2f TB/s total  %6.1f GB/s per CU  (%5.1f us per 128 KiB tile)\n", name, (long long)N, grid, nwaves, nap,
         ms, bytes / ms / 1e9, bytes / ms / 1e6 / grid, ms * 1e3 / tiles);
}
int main() {
  const int64_t M = 786432;
  char* out; (void)hipMalloc(&out, M * 2304 * 2);
  (void)hipMemset(out, 0, M * 2304 * 2);
  for (int64_t N : {768, 2304}) {
    for (int grid : {256, 64, 8}) {
      run<0, 0>("shape 0: 8 rows x 128 B", out, M, N, grid, 8, 0);
      run<3, 0>("shape 3: 4 rows x 256 B", out, M, N, grid, 8, 0);
      run<1, 0>("shape 1: 2 rows x 512 B", out, M, N, grid, 8, 0);
      run<2, 0>("shape 2: 1 KiB contiguous", out, M, N, grid, 8, 0);
      run<0, 1>("shape 0, nontemporal", out, M, N, grid, 8, 0);
      run<1, 1>("shape 1, nontemporal", out, M, N, grid, 8, 0);
      run<0, 0>("shape 0, 16 waves", out, M, N, grid, 16, 0);
      run<1, 0>("shape 1, 16 waves", out, M, N, grid, 16, 0);
      run<0, 0>("shape 0, 4 waves", out, M, N, grid, 4, 0);
    }
    run<0, 0>("shape 0, nap 8 every 2nd store", out, M, N, 256, 8, 8);
    run<0, 0>("shape 0, nap 32 every 2nd store", out, M, N, 256, 8, 32);
    run<1, 0>("shape 1, nap 32 every 2nd store", out, M, N, 256, 8, 32);
  }
  return 0;
}
