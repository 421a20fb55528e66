// How does the vector L1 (TCP) charge a wave's 4-byte gather?  Per lane, per distinct address, or per distinct cache line --
// and does it matter WHICH lanes share?  The proposal kernels of the training step run against the L1 access rate
// (profiles/r02_c2_sampler_pmc.txt); if lanes that read the same line are charged once, giving the lanes of a wave the
// SAME sample index of NEIGHBOURING rays (a camera patch: they sit in the same grid cell) instead of consecutive samples
// of one ray would cut that cost several times.
// Every lane reads table[idx[i]] 64 times (different i per round); idx is built so that groups of G lanes share an address
// (G consecutive lanes, or G lanes strided over the wave) or sit in one 128-byte line at different words.  The table is
// 2 MB (L2 resident, larger than L1).  Prints ns per wave-level load instruction and per lane.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/l1_coalesce_probe.hip -o /tmp/l1_probe && /tmp/l1_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ table, const uint32_t* __restrict__ idx,
                                                     int rounds, float* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t* my = idx + (size_t)(t >> 6) * rounds * 64 + (t & 63);  // [wave][round][lane]: the index loads are coalesced
  float acc = 0.f;
#pragma unroll 8
  for (int r = 0; r < rounds; ++r) acc += table[my[r * 64]];
  out[t] = acc;
}

int main() {
  const int n_entries = 1 << 19;  // 2 MB of floats
  const int blocks = 256 * 8, threads = blocks * 256, rounds = 64;
  std::vector<float> h_table(n_entries, 1.f);
  float *d_table, *d_out;
  uint32_t* d_idx;
  (void)hipMalloc(&d_table, n_entries * sizeof(float));
  (void)hipMalloc(&d_out, threads * sizeof(float));
  (void)hipMalloc(&d_idx, (size_t)threads * rounds * sizeof(uint32_t));
  (void)hipMemcpy(d_table, h_table.data(), n_entries * sizeof(float), hipMemcpyHostToDevice);
  struct Pat { const char* name; int G; int mode; };  // mode 0: same address, consecutive lanes; 1: same address, lanes strided;
                                                       // 2: same 128-B line, different words, consecutive lanes
  const Pat pats[] = {{"all 64 lanes distinct lines", 1, 0},        {"pairs share an address (consecutive)", 2, 0},
                      {"4 consecutive lanes share an address", 4, 0}, {"16 consecutive lanes share an address", 16, 0},
                      {"64 lanes share one address", 64, 0},         {"4 lanes share, strided by 16 lanes", 4, 1},
                      {"16 lanes share, strided by 4 lanes", 16, 1},  {"4 consecutive lanes in one line, different words", 4, 2},
                      {"16 consecutive lanes in one line, different words", 16, 2},
                      {"32 consecutive lanes in one line, different words", 32, 2}};
  std::vector<uint32_t> h_idx((size_t)threads * rounds);
  for (const Pat& p : pats) {
    srand(7);
    for (int w = 0; w < threads / 64; ++w)
      for (int r = 0; r < rounds; ++r) {
        uint32_t base[64];
        for (int k = 0; k < 64; ++k) base[k] = ((uint32_t)rand() * 2654435761u) % (uint32_t)(n_entries / 32) * 32u;  // line-aligned
        for (int lane = 0; lane < 64; ++lane) {
          int grp = p.mode == 1 ? lane % (64 / p.G) : lane / p.G;
          uint32_t a = base[grp];
          if (p.mode == 2) a += (uint32_t)(lane % p.G) % 32u;
          h_idx[((size_t)w * rounds + r) * 64 + lane] = a;
        }
      }
    (void)hipMemcpy(d_idx, h_idx.data(), h_idx.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int k = 0; k < 3; ++k) gather_kernel<<<blocks, 256>>>(d_table, d_idx, rounds, d_out);
    (void)hipEventRecord(e0);
    for (int k = 0; k < 10; ++k) gather_kernel<<<blocks, 256>>>(d_table, d_idx, rounds, d_out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_loads = 10.0 * (threads / 64) * rounds;
    printf("%-52s %8.1f us/launch  %6.2f ns per wave-load per CU-slot  (%.2f wave-loads/clk/CU at 2.4 GHz)\n", p.name, ms * 100.0,
           ms * 1e6 / wave_loads * 256, wave_loads / (ms * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
