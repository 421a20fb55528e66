// Which XCD does workgroup b of a 1-D grid run on?  The fused kernels assume b % 8 (observed dispatch order, a locality
// hint only); a design in which an XCD OWNS table levels (DESIGN §8 "what comes next" 2) needs the fact, read from the
// hardware: s_getreg_b32 HW_REG_XCC_ID.  Prints the histogram of (blockIdx.x % 8) x (xcc id) and the share of workgroups
// for which the guess holds.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/xcc_id_probe.hip -o /tmp/xcc_id_probe && /tmp/xcc_id_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void xcc_kernel(int* __restrict__ xcc_of_block, int spin) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  // keep the workgroup resident for a while so that the grid really spreads over the chip
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = (int)(id & 0xf) + (v == 12345.f ? 1 : 0);
}

int main() {
  for (int blocks : {8, 256, 2048, 16384}) {
    int* d = nullptr;
    (void)hipMalloc(&d, blocks * sizeof(int));
    xcc_kernel<<<blocks, 256>>>(d, 2000);
    std::vector<int> h(blocks);
    (void)hipMemcpy(h.data(), d, blocks * sizeof(int), hipMemcpyDeviceToHost);
    int hist[8][16] = {};
    int match = 0;
    for (int b = 0; b < blocks; ++b) {
      hist[b % 8][h[b] & 15]++;
      match += (h[b] == b % 8);
    }
    printf("grid %6d: blockIdx %% 8 == xcc id for %.1f %% of the workgroups\n", blocks, 100.0 * match / blocks);
    for (int r = 0; r < 8; ++r) {
      printf("  b%%8=%d:", r);
      for (int x = 0; x < 8; ++x) printf(" %6d", hist[r][x]);
      printf("\n");
    }
    (void)hipFree(d);
  }
  return 0;
}
