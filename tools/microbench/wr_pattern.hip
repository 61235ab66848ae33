// Store-pattern microbenchmark for the igemm epilogue (gfx950): how fast can 256-thread workgroups write a
// [ROWS][LD] bf16 matrix when every workgroup owns a (rows x piece-bytes) tile?  No loads, no LDS: pure stores.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/wr_pattern tools/microbench/wr_pattern.hip
//   run  : tools/microbench/wr_pattern [rows=200704] [ld_elems=1216]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// tile = TR rows x PB bytes; tiles_c column tiles per row block; order: column tile fastest.
// mode 0: vb = blockIdx + k * grid (round robin); mode 1: XCD-chunked (each XCD walks a contiguous range of tiles)
__global__ __launch_bounds__(256) void wr_tiles(char* __restrict__ out, long long rows, long long ld_bytes, int TR, int PB,
                                                int tiles_c, long long ntiles, int mode, int waitcnt) {
  const int lanes_per_row = PB / 16;
  const int rows_per_trip = 256 / lanes_per_row;
  const int r_in = threadIdx.x / lanes_per_row, c16 = threadIdx.x % lanes_per_row;
  const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
  long long vb0 = blockIdx.x, step = gridDim.x;
  if (mode == 1) {
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8, per = gridDim.x / 8;
    const long long chunk = (ntiles + 7) / 8;
    vb0 = xcd * chunk + slot;
    step = per;
    ntiles = (xcd + 1) * chunk < ntiles ? (xcd + 1) * chunk : ntiles;
  }
  for (long long vb = vb0; vb < ntiles; vb += step) {
    const long long rb = vb / tiles_c;
    const int cb = (int)(vb % tiles_c);
    char* base = out + rb * TR * ld_bytes + (long long)cb * PB + (long long)c16 * 16;
    const long long rmax = rows - rb * TR;
    for (int r = r_in; r < TR; r += rows_per_trip)
      if (r < rmax && (long long)cb * PB + c16 * 16 + 16 <= ld_bytes) *(uint4*)(base + r * ld_bytes) = v;
    if (waitcnt) { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); }
  }
}

int main(int argc, char** argv) {
  const long long rows = argc > 1 ? atoll(argv[1]) : 200704;
  const long long ld = argc > 2 ? atoll(argv[2]) : 1216;
  const long long ld_bytes = ld * 2, total = rows * ld_bytes;
  char* out;
  CK(hipMalloc(&out, total + 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Cfg { const char* name; int TR, PB; };
  const int fullrow = (int)ld_bytes;
  std::vector<Cfg> cfgs = {{"128r x 256B", 128, 256}, {"64r x 512B", 64, 512}, {"32r x 1024B", 32, 1024}, {"256r x 128B", 256, 128},
                           {"128r x 128B", 128, 128}, {"128r x 512B", 128, 512}, {"256r x 256B", 256, 256}};
  printf("matrix %lld x %lld bf16 = %.1f MB\n", rows, ld, total / 1e6);
  // linear reference
  {
    const int PB = 4096, TR = 8;   // 32 KB contiguous per tile when ld_bytes == PB: emulate by treating memory as [total/4096][4096]
    const long long r2 = total / 4096, nt = (r2 + TR - 1) / TR;
    for (int grid : {512, 1024, 2048}) {
      wr_tiles<<<grid, 256>>>(out, r2, 4096, TR, PB, 1, nt, 0, 0);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) wr_tiles<<<grid, 256>>>(out, r2, 4096, TR, PB, 1, nt, 0, 0);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-16s grid %5d mode 0 wait 0 : %8.1f us  %6.2f TB/s\n", "linear 32KB", grid, ms * 100, total / (ms / 10 * 1e-3) / 1e12);
    }
  }
  for (const Cfg& c : cfgs) {
    if (c.PB % 16 || 256 % (c.PB / 16)) continue;
    const int tiles_c = (int)((ld_bytes + c.PB - 1) / c.PB);
    const long long nt = ((rows + c.TR - 1) / c.TR) * tiles_c;
    for (int grid : {512, 1024})
      for (int mode : {0, 1})
        for (int wait : {0, 1}) {
          wr_tiles<<<grid, 256>>>(out, rows, ld_bytes, c.TR, c.PB, tiles_c, nt, mode, wait);
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0));
          for (int i = 0; i < 10; ++i) wr_tiles<<<grid, 256>>>(out, rows, ld_bytes, c.TR, c.PB, tiles_c, nt, mode, wait);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          printf("%-16s grid %5d mode %d wait %d : %8.1f us  %6.2f TB/s\n", c.name, grid, mode, wait, ms * 100,
                 total / (ms / 10 * 1e-3) / 1e12);
        }
  }
  (void)fullrow;
  return 0;
}
