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
#include <mutex>
#include <thread>
#include <vector>

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
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((0u - (v & 1u)) & 0x9908b0dfu);   // branch-free
  }
  inline void next_state() {
    left = MT_N; next = 0;
    // same recurrence as ATen's mt19937, written as index loops the vectoriser accepts (dependence distance 227)
    for (int i = 0; i < MT_N - MT_M; ++i) st[i] = st[i + MT_M] ^ twist(st[i], st[i + 1]);
    for (int i = MT_N - MT_M; i < MT_N - 1; ++i) st[i] = st[i + MT_M - MT_N] ^ twist(st[i], st[i + 1]);
    st[MT_N - 1] = st[MT_M - 1] ^ twist(st[MT_N - 1], st[0]);
  }
  // advance by n draws without producing them (a draw = `if (--left == 0) next_state(); st[next++]`)
  inline void skip(uint64_t n) {
    while (n) {
      const uint64_t avail = (uint64_t)(left - 1);
      if (n <= avail) { left -= (int)n; next += (uint32_t)n; return; }
      n -= avail;
      next_state(); next = 1; n -= 1;          // the draw that regenerates consumes st[0]; left stays MT_N
    }
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


namespace {
struct GridGeom { long edge, side, per_block, take, last, nblk, per_row; };
inline bool grid_geom(long high, long shape, int cut, int mirror, GridGeom& q) {
  q.edge = lround(sqrt((double)high));            // python round(): sqrt(int) is never exactly x.5, so they agree
  q.side = q.edge / cut;
  if (q.side <= 1) return false;                  // the reference falls back to the 1-D sampler, nothing drawn
  q.per_block = shape * q.edge * q.edge / high / ((long)cut * cut);
  q.take = mirror ? q.per_block / 2 : q.per_block;
  q.last = q.edge - (long)(cut - 1) * q.side;
  q.nblk = (long)cut * cut;
  q.per_row = mirror ? 2 * q.take : q.take;
  return true;
}
// number of generator draws of one call when it does not depend on the drawn values: no candidate can be
// dropped by the `< high` filter (edge^2 <= high, exact float32 round trip) -> kept = nblk * per_row
inline bool grid_draws(long high, long shape, int cut, int mirror, uint64_t* draws) {
  GridGeom q;
  if (!grid_geom(high, shape, cut, mirror, q)) return false;
  if (q.edge * q.edge > high || high > (1l << 24)) return false;
  uint64_t d = 0;
  for (int bi = 0; bi < cut; ++bi) {
    const long h = bi == cut - 1 ? q.last : q.side;
    for (int bj = 0; bj < cut; ++bj) { const long w = bj == cut - 1 ? q.last : q.side; d += (uint64_t)(h * w - 1 + q.take); }
  }
  const long kept = q.nblk * q.per_row;
  if (kept > 0) d += (uint64_t)(kept - 1);
  if (kept < shape) d += (uint64_t)(shape - kept);
  *draws = d;
  return true;
}
// one sampler call on generator g (loss_helper_3d.py:120-184 / :187-268); returns shape, or 0 for the fallback
inline size_t grid_scratch_len(const GridGeom& q) { return (size_t)(2 * (q.nblk * q.per_row + 1) + q.last * q.last + 1); }
// scratch: grid_scratch_len() int64s (no allocation in here: worker threads would serialise on the mm lock)
long grid_sample_mt(MT& g, long high, long shape, int cut, int mirror, int64_t* out, int64_t* scratch) {
  GridGeom q;
  if (!grid_geom(high, shape, cut, mirror, q)) return 0;
  const long edge = q.edge, side = q.side, take = q.take, last = q.last, nblk = q.nblk, per_row = q.per_row;
  int64_t* vals = scratch;
  int64_t* shuf = scratch + (nblk * per_row + 1);
  int64_t* perm = shuf + (nblk * per_row + 1);
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
  randperm(g, kept, shuf);
  long m = kept < shape ? kept : shape;
  for (long i = 0; i < m; ++i) out[i] = vals[shuf[i]];
  for (long i = kept; i < shape; ++i) out[i] = (int64_t)(g() % (uint32_t)high);   // one draw per padded element
  return shape;
}
// process-lifetime scratch arena (pages stay resident from call to call); one sampler sequence at a time
static std::vector<int64_t> g_arena;
static std::mutex g_arena_mutex;
inline bool load_state(const uint8_t* state, long state_bytes, MT& g) {
  if (state_bytes != (long)sizeof(TorchCpuState)) return false;
  const TorchCpuState* ts = reinterpret_cast<const TorchCpuState*>(state);
  if (!ts->legacy.seeded) return false;
  for (int i = 0; i < MT_N; ++i) g.st[i] = (uint32_t)ts->legacy.state[i];
  g.left = ts->legacy.left; g.next = (uint32_t)ts->legacy.next;
  return true;
}
inline void store_state(uint8_t* state, const MT& g) {
  TorchCpuState* ts = reinterpret_cast<TorchCpuState*>(state);
  for (int i = 0; i < MT_N; ++i) ts->legacy.state[i] = g.st[i];
  ts->legacy.left = g.left; ts->legacy.next = g.next;
}
}  // namespace

extern "C" {

// returns shape on success; 0 when the reference falls back to the 1-D sampler (edge//cut <= 1,
// nothing drawn); <0 on error.  `state` = torch.get_rng_state() bytes, updated in place.
long arco_grid_sample(uint8_t* state, long state_bytes, long high, long shape, int cut, int mirror, int64_t* out) {
  if (high <= 0 || shape <= 0 || cut <= 0) return -1;
  if (high >= (1l << 31)) return -3;                       // 32-bit draw path only (always true for pixel counts)
  MT g;
  if (!load_state(state, state_bytes, g)) return state_bytes != (long)sizeof(TorchCpuState) ? -1 : -2;
  GridGeom q;
  if (!grid_geom(high, shape, cut, mirror, q)) return 0;
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  if (g_arena.size() < grid_scratch_len(q)) g_arena.resize(grid_scratch_len(q));
  const long rc = grid_sample_mt(g, high, shape, cut, mirror, out, g_arena.data());
  if (rc > 0) store_state(state, g);
  return rc;
}

// A SEQUENCE of sampler calls in generator order (one step draws anchors and negatives for every valid class:
// 2C calls, ~1 M draws at config 2).  The calls are inherently ordered through the generator, but a call whose
// number of draws does not depend on the drawn values (grid_draws) can run on a COPY of the state in a worker
// thread while this thread skips the generator ahead by that many draws (state regeneration only, vectorised)
// and goes on with the next call.  Stops at the first call that needs the reference's 1-D fallback (which uses
// python's `random`): returns its index (n_jobs when all ran); jobs before it are complete and `state` is the
// generator state right before it.  <0 on error.
long arco_grid_sample_many(uint8_t* state, long state_bytes, int n_jobs, const long* highs, const long* shapes, int cut,
                           int mirror, int64_t* const* outs, int max_threads) {
  if (n_jobs < 0 || cut <= 0) return -1;
  MT g;
  if (!load_state(state, state_bytes, g)) return state_bytes != (long)sizeof(TorchCpuState) ? -1 : -2;
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  std::vector<size_t> off(n_jobs + 1, 0);
  for (int j = 0; j < n_jobs; ++j) {
    GridGeom q;
    const bool ok = highs[j] > 0 && shapes[j] > 0 && grid_geom(highs[j], shapes[j], cut, mirror, q);
    off[j + 1] = off[j] + (ok ? grid_scratch_len(q) : 0);
  }
  if (g_arena.size() < off[n_jobs]) g_arena.resize(off[n_jobs]);
  std::vector<std::thread> workers;
  std::vector<MT> copies((size_t)n_jobs);
  long done = n_jobs;
  for (int j = 0; j < n_jobs; ++j) {
    const long high = highs[j], shape = shapes[j];
    if (high <= 0 || shape <= 0 || high >= (1l << 31)) { done = -1; break; }
    int64_t* scratch = g_arena.data() + off[j];
    uint64_t draws = 0;
    if (shape >= 8192 && (int)workers.size() < max_threads && grid_draws(high, shape, cut, mirror, &draws)) {
      copies[j] = g;
      MT* copy = &copies[j];
      int64_t* out = outs[j];
      workers.emplace_back([copy, high, shape, cut, mirror, out, scratch]() { grid_sample_mt(*copy, high, shape, cut, mirror, out, scratch); });
      g.skip(draws);
      continue;
    }
    if (grid_sample_mt(g, high, shape, cut, mirror, outs[j], scratch) == 0) { done = j; break; }
  }
  for (auto& t : workers) t.join();
  if (done >= 0) store_state(state, g);
  return done;
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
