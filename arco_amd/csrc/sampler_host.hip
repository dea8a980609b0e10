// Host-side (CPU) native replay of the reference's stratified grid samplers
// (loss_helper_3d.py:120-184 smc, :187-268 asmc) on torch's CPU generator state.
//
// Bit-exact sample indices are defined by the torch CPU default generator call sequence
// (per block randperm(n) then randint(n,(k,)), one randperm shuffle, one randint per padded
// element).  torch.randperm / torch.randint on CPU are serial loops over a 32-bit mt19937
// (ATen CPUGeneratorImpl): randperm(n): r[i]=i; for i<n-1: z = rnd32() % (n-i); swap(r[i], r[i+z]);
// randint(high): rnd32() % high  (high < 2^32).  This file re-implements exactly that on the
// generator's serialized state (torch.get_rng_state(): CPUGeneratorImplState, 5056 bytes), so the
// Python side does  get_rng_state -> arco_grid_sample -> set_rng_state  and the generator ends
// in the same state as if torch had made the calls.  ~100x faster than issuing the torch calls
// (no per-call tensor allocation / dispatch), which takes the sampler off the step's critical path.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

namespace {
constexpr int MT_N = 624, MT_M = 397;
struct LegacyState {          // at::CPUGeneratorImplStateLegacy
  uint64_t the_initial_seed; int left; int seeded; uint64_t next; uint64_t state[MT_N];
  double normal_x, normal_y, normal_rho; int normal_is_valid;
};
struct TorchCpuState { LegacyState legacy; float next_float_normal_sample; bool is_next_float_normal_sample_valid; };
static_assert(sizeof(TorchCpuState) == 5056, "torch CPU generator state layout");

struct MT {
  uint32_t st[MT_N]; int left; uint32_t next;
  static inline uint32_t twist(uint32_t u, uint32_t v) {
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
  }
  inline void next_state() {
    uint32_t* p = st; left = MT_N; next = 0;
    for (int j = MT_N - MT_M + 1; --j; p++) *p = p[MT_M] ^ twist(p[0], p[1]);
    for (int j = MT_M; --j; p++) *p = p[MT_M - MT_N] ^ twist(p[0], p[1]);
    *p = p[MT_M - MT_N] ^ twist(p[0], st[0]);
  }
  inline uint32_t operator()() {
    if (--left == 0) next_state();
    uint32_t y = st[next++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
  }
};
inline void randperm(MT& g, int64_t n, int64_t* r) {
  for (int64_t i = 0; i < n; ++i) r[i] = i;
  for (int64_t i = 0; i < n - 1; ++i) { const int64_t z = (int64_t)(g() % (uint32_t)(n - i)); const int64_t t = r[i]; r[i] = r[z + i]; r[z + i] = t; }
}
}  // namespace

extern "C" {

// returns shape on success; 0 when the reference falls back to the 1-D sampler (edge//cut <= 1,
// nothing drawn); <0 on error.  `state` = torch.get_rng_state() bytes, updated in place.
long arco_grid_sample(uint8_t* state, long state_bytes, long high, long shape, int cut, int mirror, int64_t* out) {
  if (state_bytes != (long)sizeof(TorchCpuState) || high <= 0 || shape <= 0 || cut <= 0) return -1;
  if (high >= (1l << 31)) return -3;                       // 32-bit draw path only (always true for pixel counts)
  const long edge = lround(sqrt((double)high));             // python round(): ties-to-even vs lround half-away:
  {                                                         // sqrt(int) is never exactly x.5, so they agree
  }
  const long side = edge / cut;
  if (side <= 1) return 0;
  TorchCpuState* ts = reinterpret_cast<TorchCpuState*>(state);
  if (!ts->legacy.seeded) return -2;
  MT g;
  for (int i = 0; i < MT_N; ++i) g.st[i] = (uint32_t)ts->legacy.state[i];
  g.left = ts->legacy.left; g.next = (uint32_t)ts->legacy.next;

  const long per_block = shape * edge * edge / high / ((long)cut * cut);
  const long take = mirror ? per_block / 2 : per_block;
  const long last = edge - (long)(cut - 1) * side;
  const long nblk = (long)cut * cut;
  const long per_row = mirror ? 2 * take : take;
  int64_t* vals = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nblk * per_row + 1));
  int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)(last * last + 1));
  long b = 0;
  for (int bi = 0; bi < cut; ++bi) {
    const long h = bi == cut - 1 ? last : side;
    for (int bj = 0; bj < cut; ++bj, ++b) {
      const long w = bj == cut - 1 ? last : side;
      const long n = h * w;
      randperm(g, n, perm);
      const long org = (bi * side) * edge + bj * side;
      const long fin = org + (bi * side + h - 1) * edge + bj * side + w - 1;   // first + last == int64(2*mean(block))
      int64_t* row = vals + b * per_row;
      for (long t = 0; t < take; ++t) {
        const int64_t loc = perm[g() % (uint32_t)n];
        const int64_t v = org + (loc / w) * edge + loc % w;
        row[t] = v;
        if (mirror) row[take + t] = fin - v;
      }
    }
  }
  // float32 round trip (torch.Tensor(...).long(), :163 / :245-246), keep < high
  long kept = 0;
  for (long i = 0; i < nblk * per_row; ++i) {
    const int64_t v = (int64_t)(float)vals[i];
    if (v < high) vals[kept++] = v;
  }
  int64_t* shuf = (int64_t*)malloc(sizeof(int64_t) * (size_t)(kept + 1));
  randperm(g, kept, shuf);
  long m = kept < shape ? kept : shape;
  for (long i = 0; i < m; ++i) out[i] = vals[shuf[i]];
  for (long i = kept; i < shape; ++i) out[i] = (int64_t)(g() % (uint32_t)high);   // one draw per padded element
  free(vals); free(perm); free(shuf);
  for (int i = 0; i < MT_N; ++i) ts->legacy.state[i] = g.st[i];
  ts->legacy.left = g.left; ts->legacy.next = g.next;
  return shape;
}

// plain torch.randint(high, (n,)) replay (func not in {'asmc','smc'}, and the high < 16 fallbacks)
long arco_randint(uint8_t* state, long state_bytes, long high, long n, int64_t* out) {
  if (state_bytes != (long)sizeof(TorchCpuState) || high <= 0 || high >= (1l << 31)) return -1;
  TorchCpuState* ts = reinterpret_cast<TorchCpuState*>(state);
  MT g;
  for (int i = 0; i < MT_N; ++i) g.st[i] = (uint32_t)ts->legacy.state[i];
  g.left = ts->legacy.left; g.next = (uint32_t)ts->legacy.next;
  for (long i = 0; i < n; ++i) out[i] = (int64_t)(g() % (uint32_t)high);
  for (int i = 0; i < MT_N; ++i) ts->legacy.state[i] = g.st[i];
  ts->legacy.left = g.left; ts->legacy.next = g.next;
  return n;
}

}  // extern "C"
