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
#include <pthread.h>
#include <thread>
#include <vector>
#include <chrono>
#include <stdio.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <unistd.h>

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
  // STREAM mode (arco_mt_pregen): the generator's future state blocks were computed ahead of time - while the host was
  // waiting for the GPU's counters anyway - and sit in `stream` (block b = the state after the b-th regeneration from the
  // base state).  Moving to the next block is then a pointer step instead of 624 twists, so skipping ahead is O(1) and
  // copies of the generator positioned anywhere in the stream can run in parallel.  blk = -1: still in the base block
  // (st).  Past the last pregenerated block the generator carries on by itself (st <- last block, plain mode).
  const uint32_t* stream = nullptr; long blk = -1, nblk = 0;
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
  inline const uint32_t* cur() const { return (stream && blk >= 0) ? stream + blk * MT_N : st; }
  inline void advance() {                      // what next_state() is in plain mode
    if (stream && blk + 1 < nblk) { ++blk; left = MT_N; next = 0; return; }
    if (stream) { if (blk >= 0) memcpy(st, stream + blk * MT_N, sizeof(st)); stream = nullptr; blk = -1; }
    next_state();
  }
  // advance by n draws without producing them (a draw = `if (--left == 0) advance(); cur()[next++]`)
  inline void skip(uint64_t n) {
    while (n) {
      const uint64_t avail = (uint64_t)(left - 1);
      if (n <= avail) { left -= (int)n; next += (uint32_t)n; return; }
      n -= avail;
      if (stream && blk + 1 < nblk) {          // whole pregenerated blocks at once
        const uint64_t whole = (n - 1) / MT_N;  // the draw that enters a block consumes its word 0; left stays MT_N
        const uint64_t room = (uint64_t)(nblk - 1 - blk) - 1;
        const uint64_t hop = whole < room ? whole : room;
        blk += (long)hop; n -= hop * MT_N;
      }
      advance(); next = 1; n -= 1;
    }
  }
  // draws that can still be served without leaving the pregenerated blocks (0 in plain mode)
  inline uint64_t stream_room() const {
    if (!stream) return 0;
    return (uint64_t)(left - 1) + (uint64_t)(nblk - 1 - blk) * MT_N;
  }
  inline uint32_t operator()() {
    if (--left == 0) advance();
    uint32_t y = cur()[next++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
  }
  inline void materialize() {                  // leave stream mode keeping the position (before the state is stored)
    if (stream && blk >= 0) memcpy(st, stream + blk * MT_N, sizeof(st));
    stream = nullptr; blk = -1;
  }
};
// (32-bit permutation arrays: the swaps are random accesses over the whole array - a 142 000-element block of a 3-D anchor
// call or the 131 072-element final shuffle of a negative call is 0.5 MB instead of 1.1 MB, inside the core's L2)
template <typename T>
inline void randperm(MT& g, int64_t n, T* r) {
  for (int64_t i = 0; i < n; ++i) r[i] = (T)i;
  for (int64_t i = 0; i < n - 1; ++i) { const int64_t z = (int64_t)(g() % (uint32_t)(n - i)); const T t = r[i]; r[i] = r[z + i]; r[z + i] = t; }
}
}  // namespace


namespace {
// Persistent worker threads for the sampler replay (creating ~20 std::threads per step cost 0.05-0.1 ms EACH under
// contention, more than the work they were given).  Tasks may submit tasks; a thread that waits for a group runs queued
// tasks meanwhile.  The pool is leaked on purpose (no join at exit) and rebuilt in a forked child.
class Pool {
  std::vector<std::thread> th; std::deque<std::function<void()>> q; std::mutex m; std::condition_variable cv;
  void loop() {
    for (;;) {
      std::function<void()> f;
      { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return !q.empty(); }); f = std::move(q.front()); q.pop_front(); }
      f();
    }
  }
 public:
  explicit Pool(int n) { for (int i = 0; i < n; ++i) { th.emplace_back([this] { loop(); }); th.back().detach(); } }
  void submit(std::function<void()> f) { { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(f)); } cv.notify_one(); }
  bool try_run_one() {
    std::function<void()> f;
    { std::lock_guard<std::mutex> lk(m); if (q.empty()) return false; f = std::move(q.front()); q.pop_front(); }
    f();
    return true;
  }
};
// fork(): the child inherits counters of tasks whose threads do not exist there (a deferred sampler sequence or a
// pregeneration in flight would make its first Group::wait spin forever) and possibly locked mutexes - the atfork child
// handler (after_fork_child, below the globals it resets) clears them; the pool itself is rebuilt on first use.
static void after_fork_child();
static std::mutex g_pool_mutex;
static Pool* g_pool = nullptr;
inline Pool& pool() {
  static pid_t owner = 0; static bool hooked = false;
  std::lock_guard<std::mutex> lk(g_pool_mutex);
  if (!hooked) { pthread_atfork(nullptr, nullptr, after_fork_child); hooked = true; }
  if (!g_pool || owner != getpid()) { g_pool = new Pool(16); owner = getpid(); }
  return *g_pool;
}
struct Group {
  std::atomic<int> pending{0};
  void run(std::function<void()> f) { pending.fetch_add(1); pool().submit([this, f]() { f(); pending.fetch_sub(1); }); }
  void wait() { Pool& p = pool(); while (pending.load() > 0) { if (!p.try_run_one()) std::this_thread::yield(); } }
  ~Group() { wait(); }
};

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
// number of generator draws of one call when it does not depend on the drawn values.  Two cases:
//  (a) no candidate can be dropped by the `< high` filter (edge^2 <= high, exact float32 round trip): kept = nblk * per_row;
//  (b) candidates may be dropped (edge^2 > high: the anchor calls - high = the class's candidate count is rarely a square), but at most
//      `shape` are drawn (nblk * per_row <= shape): the final randperm(kept') takes kept' - 1 draws and the padding shape - kept' draws -
//      shape - 1 together WHATEVER kept' is (kept' >= 1: block (0, 0) holds values < high only, and its mirror images stay inside the block).
// *kept_fixed: case (a) (the caller may start the final shuffle at a known stream offset).
inline bool grid_draws(long high, long shape, int cut, int mirror, uint64_t* draws, bool* kept_fixed = nullptr) {
  GridGeom q;
  if (!grid_geom(high, shape, cut, mirror, q)) return false;
  if (q.edge * q.edge > (1l << 24) || high > (1l << 24)) return false;          // (float32 round trip of the picked values must be exact)
  const long kept = q.nblk * q.per_row;
  const bool fixed = q.edge * q.edge <= high;
  if (!fixed && !(kept <= shape && q.take >= 1 && kept >= 1)) return false;
  uint64_t d = 0;
  for (int bi = 0; bi < cut; ++bi) {
    const long h = bi == cut - 1 ? q.last : q.side;
    for (int bj = 0; bj < cut; ++bj) { const long w = bj == cut - 1 ? q.last : q.side; d += (uint64_t)(h * w - 1 + q.take); }
  }
  if (kept > 0) d += (uint64_t)(kept - 1);
  if (kept < shape) d += (uint64_t)(shape - kept);
  *draws = d;
  if (kept_fixed) *kept_fixed = fixed;
  return true;
}
// one sampler call on generator g (loss_helper_3d.py:120-184 / :187-268); returns shape, or 0 for the fallback
inline size_t grid_scratch_len(const GridGeom& q) { return (size_t)(2 * (q.nblk * q.per_row + 1) + q.nblk * (q.last * q.last + 1)); }
// scratch: grid_scratch_len() int64s (no allocation in here: worker threads would serialise on the mm lock)
// one grid block (bi, bj) of a call: randperm(h * w), then `take` picks; consumes exactly h * w - 1 + take draws
inline void grid_block(MT& g, const GridGeom& q, int cut, int mirror, int bi, int bj, int32_t* perm, int64_t* row) {
  const long h = bi == cut - 1 ? q.last : q.side, w = bj == cut - 1 ? q.last : q.side, n = h * w;
  randperm(g, n, perm);
  const long org = (bi * q.side) * q.edge + bj * q.side;
  const long fin = org + (bi * q.side + h - 1) * q.edge + bj * q.side + w - 1;   // first + last == int64(2*mean(block))
  // x % n, loc / w, loc % w with the block's invariant divisors as multiplications (Lemire's exact 32-bit fastmod /
  // fastdiv; n, w >= 2): the negative calls make 8192 picks per block, three hardware divisions each
  const uint64_t Mn = ~0ull / (uint64_t)n + 1, Mw = ~0ull / (uint64_t)w + 1;
  for (long t = 0; t < q.take; ++t) {
    const uint32_t x = g();
    const uint32_t idx = (uint32_t)(((__uint128_t)(Mn * x) * (uint64_t)n) >> 64);      // == x % n
    const int64_t loc = perm[idx];
    const int64_t qd = (int64_t)(((__uint128_t)Mw * (uint64_t)loc) >> 64);             // == loc / w
    const int64_t v = org + qd * q.edge + (loc - qd * w);
    row[t] = v;
    if (mirror) row[q.take + t] = fin - v;
  }
}
// par_threads > 1 and the generator in stream mode with every block's draws pregenerated: the blocks (whose draw counts
// are fixed by the geometry) run on copies of the generator positioned at their offsets, par_threads at a time - the
// anchor call of a class with ~600 000 candidates is 16 permutations of ~36 000 elements, 1.4 ms in one thread
// fixed_kept (the caller checked grid_draws(): no candidate can be dropped, kept == nblk * per_row): the final
// randperm(kept) - a third of a negative call's time, inherently serial - starts at a known stream offset too and runs in
// its own thread beside the blocks.
long grid_sample_mt(MT& g, long high, long shape, int cut, int mirror, int64_t* out, int64_t* scratch, int par_threads = 1,
                    bool fixed_kept = false) {
  GridGeom q;
  if (!grid_geom(high, shape, cut, mirror, q)) return 0;
  const long nblk = q.nblk, per_row = q.per_row;
  int64_t* vals = scratch;
  int32_t* shuf = reinterpret_cast<int32_t*>(scratch + (nblk * per_row + 1));        // (int32 views of the int64 arena)
  int32_t* perm0 = reinterpret_cast<int32_t*>(scratch + 2 * (nblk * per_row + 1));
  const long perm_len = q.last * q.last + 1;
  uint64_t block_draws = 0;
  for (int bi = 0; bi < cut; ++bi)
    for (int bj = 0; bj < cut; ++bj)
      block_draws += (uint64_t)((bi == cut - 1 ? q.last : q.side) * (bj == cut - 1 ? q.last : q.side) - 1 + q.take);
  static const bool trace = getenv("ARCO_SAMPLER_TRACE") != nullptr;
  auto T0 = std::chrono::steady_clock::now();
  auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count(); };
  Group shuffle_group;
  bool shuffled = false;
  const long kept_fixed = nblk * per_row;
  if (fixed_kept && par_threads > 1 && kept_fixed >= 32768 && g.stream_room() >= block_draws + (uint64_t)kept_fixed) {
    MT g2 = g; g2.skip(block_draws);
    shuffle_group.run([g2, kept_fixed, shuf]() mutable { randperm(g2, kept_fixed, shuf); });
    shuffled = true;
  }
  if (par_threads > 1 && block_draws >= 65536 && g.stream_room() >= block_draws) {
    std::vector<MT> gs((size_t)nblk);
    uint64_t off = 0;
    for (long b = 0; b < nblk; ++b) {
      const int bi = (int)(b / cut), bj = (int)(b % cut);
      gs[b] = g; gs[b].skip(off);
      off += (uint64_t)((bi == cut - 1 ? q.last : q.side) * (bj == cut - 1 ? q.last : q.side) - 1 + q.take);
    }
    const int nt = par_threads < (int)nblk ? par_threads : (int)nblk;
    Group blocks;
    for (int t = 1; t < nt; ++t)
      blocks.run([&, t]() { for (long b = t; b < nblk; b += nt) grid_block(gs[b], q, cut, mirror, (int)(b / cut), (int)(b % cut), perm0 + b * perm_len, vals + b * per_row); });
    for (long b = 0; b < nblk; b += nt) grid_block(gs[b], q, cut, mirror, (int)(b / cut), (int)(b % cut), perm0 + b * perm_len, vals + b * per_row);
    blocks.wait();
    g.skip(block_draws);
  } else {
    long b = 0;
    for (int bi = 0; bi < cut; ++bi)
      for (int bj = 0; bj < cut; ++bj, ++b) grid_block(g, q, cut, mirror, bi, bj, perm0, vals + b * per_row);
  }
  const double t_blocks = ms();
  // float32 round trip (torch.Tensor(...).long(), :163 / :245-246), keep < high
  long kept = 0;
  for (long i = 0; i < nblk * per_row; ++i) {
    const int64_t v = (int64_t)(float)vals[i];
    if (v < high) vals[kept++] = v;
  }
  const double t_filter = ms();
  if (shuffled) { shuffle_group.wait(); g.skip(kept > 1 ? (uint64_t)(kept - 1) : 0); }   // (kept == kept_fixed here)
  else randperm(g, kept, shuf);
  const double t_shuffle = ms();
  long m = kept < shape ? kept : shape;
  for (long i = 0; i < m; ++i) out[i] = vals[shuf[i]];
  if (trace && shape >= 32768) fprintf(stderr, "[sampler]   call high %ld: blocks %.3f filter %.3f shuffle(join) %.3f gather %.3f ms\n", high, t_blocks, t_filter - t_blocks, t_shuffle - t_filter, ms() - t_shuffle);
  for (long i = kept; i < shape; ++i) out[i] = (int64_t)(g() % (uint32_t)high);   // one draw per padded element
  return shape;
}
// process-lifetime scratch arena (pages stay resident from call to call); one sampler sequence at a time
static std::vector<int64_t> g_arena;
static std::mutex g_arena_mutex;
// a deferred sampler sequence (arco_grid_sample_many_async) whose worker calls may still be running; owns the arena
static Group g_many_workers;
static std::vector<MT> g_many_copies;
static bool g_many_active = false;
static void many_finish() { if (g_many_active) { g_many_workers.wait(); g_many_active = false; } }
inline bool load_state(const uint8_t* state, long state_bytes, MT& g) {
  if (state_bytes != (long)sizeof(TorchCpuState)) return false;
  const TorchCpuState* ts = reinterpret_cast<const TorchCpuState*>(state);
  if (!ts->legacy.seeded) return false;
  for (int i = 0; i < MT_N; ++i) g.st[i] = (uint32_t)ts->legacy.state[i];
  g.left = ts->legacy.left; g.next = (uint32_t)ts->legacy.next;
  return true;
}
inline void store_state(uint8_t* state, const MT& g_in) {
  MT g = g_in; g.materialize();
  TorchCpuState* ts = reinterpret_cast<TorchCpuState*>(state);
  for (int i = 0; i < MT_N; ++i) ts->legacy.state[i] = g.st[i];
  ts->legacy.left = g.left; ts->legacy.next = g.next;
}
// ---- pregenerated state blocks (arco_mt_pregen): valid for the generator state they were made from
static std::vector<uint32_t> g_stream;
static MT g_stream_base;
static long g_stream_blocks = 0;
static Group g_stream_group;
static std::mutex g_stream_mutex;
inline void stream_join() { g_stream_group.wait(); }
// switch g to stream mode when it is exactly the state the blocks were generated from
inline void stream_attach(MT& g) {
  std::lock_guard<std::mutex> lk(g_stream_mutex);
  stream_join();
  if (g_stream_blocks <= 0 || g.left != g_stream_base.left || g.next != g_stream_base.next ||
      memcmp(g.st, g_stream_base.st, sizeof(g.st)) != 0) return;
  g.stream = g_stream.data(); g.blk = -1; g.nblk = g_stream_blocks;
}
// child side of fork(): no worker thread survived - forget every in-flight task and pregenerated block, re-create the
// mutexes (one may have been held by a thread that no longer exists), let pool() build a fresh pool on first use
static void after_fork_child() {
  new (&g_pool_mutex) std::mutex(); new (&g_arena_mutex) std::mutex(); new (&g_stream_mutex) std::mutex();
  g_pool = nullptr;
  g_many_workers.pending.store(0); g_many_active = false; g_many_copies.clear();
  g_stream_group.pending.store(0); g_stream_blocks = 0;
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
  many_finish();
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
// deferred form: the calls that run in worker threads (the negative draws: their number of draws is known before they
// run) need not have FINISHED for the generator's final state to be known - arco_grid_sample_many_async returns as soon as
// the inline calls are done and the last worker is launched (state stored), the trainer draws its next random numbers
// (the equivariance warp) and queues ~5 ms of GPU work, and collects the indices with arco_grid_sample_many_finish()
// before it uploads them.  Same outputs, same generator state as arco_grid_sample_many.
static long sample_many_impl(uint8_t* state, long state_bytes, int n_jobs, const long* highs, const long* shapes, int cut,
                             int mirror, int64_t* const* outs, int max_threads, bool deferred);
void arco_grid_sample_many_finish() { std::lock_guard<std::mutex> lock(g_arena_mutex); many_finish(); }
long arco_grid_sample_many_async(uint8_t* state, long state_bytes, int n_jobs, const long* highs, const long* shapes, int cut,
                                 int mirror, int64_t* const* outs, int max_threads) {
  return sample_many_impl(state, state_bytes, n_jobs, highs, shapes, cut, mirror, outs, max_threads, true);
}
long arco_grid_sample_many(uint8_t* state, long state_bytes, int n_jobs, const long* highs, const long* shapes, int cut,
                           int mirror, int64_t* const* outs, int max_threads) {
  return sample_many_impl(state, state_bytes, n_jobs, highs, shapes, cut, mirror, outs, max_threads, false);
}
static long sample_many_impl(uint8_t* state, long state_bytes, int n_jobs, const long* highs, const long* shapes, int cut,
                             int mirror, int64_t* const* outs, int max_threads, bool deferred) {
  if (n_jobs < 0 || cut <= 0) return -1;
  MT g;
  if (!load_state(state, state_bytes, g)) return state_bytes != (long)sizeof(TorchCpuState) ? -1 : -2;
  static const bool trace = getenv("ARCO_SAMPLER_TRACE") != nullptr;
  auto T0 = std::chrono::steady_clock::now();
  auto ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count(); };
  stream_attach(g);                    // pregenerated blocks of exactly this state, if any (arco_mt_pregen)
  if (trace) fprintf(stderr, "[sampler] attach %.3f ms (stream %d)\n", ms(), g.stream != nullptr);
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  many_finish();                       // (a deferred sequence still in flight owns the arena)
  std::vector<size_t> off(n_jobs + 1, 0);
  for (int j = 0; j < n_jobs; ++j) {
    GridGeom q;
    const bool ok = highs[j] > 0 && shapes[j] > 0 && grid_geom(highs[j], shapes[j], cut, mirror, q);
    off[j + 1] = off[j] + (ok ? grid_scratch_len(q) : 0);
  }
  if (g_arena.size() < off[n_jobs]) g_arena.resize(off[n_jobs]);
  Group& workers = g_many_workers; int n_workers = 0;
  std::vector<MT>& copies = g_many_copies;
  copies.assign((size_t)n_jobs, MT());
  g_many_active = true;
  long done = n_jobs;
  for (int j = 0; j < n_jobs; ++j) {
    const long high = highs[j], shape = shapes[j];
    if (high <= 0 || shape <= 0 || high >= (1l << 31)) { done = -1; break; }
    int64_t* scratch = g_arena.data() + off[j];
    {   // the 1-D fallback's first branch (loss_helper.py:207-208 / :255-256: `high // patch > shape or high < patch` ->
        // torch.randint(high, size=(shape,)), one 32-bit draw per element) natively: with C = 2 the banks of a volume step keep their
        // single initial row, so every negative call is randint(1, (Q * Nn,)) - handled in Python it cut the sequence (and joined
        // the deferred anchor calls in front of it) at every class
      GridGeom qq;
      if (!grid_geom(high, shape, cut, mirror, qq) && (high / 16 > shape || high < 16)) {
        int64_t* out = outs[j];
        if (shape >= 8192 && n_workers < max_threads) {
          copies[j] = g;
          MT* copy = &copies[j];
          workers.run([copy, high, shape, out]() { for (long i = 0; i < shape; ++i) out[i] = (int64_t)((*copy)() % (uint32_t)high); });
          ++n_workers;
          g.skip((uint64_t)shape);
        } else {
          for (long i = 0; i < shape; ++i) out[i] = (int64_t)(g() % (uint32_t)high);
        }
        continue;
      }
    }
    uint64_t draws = 0; bool kfix = false;
    // worker calls: the negative draws (large shape) and - new in round 3 - the anchor calls of classes with many candidates (the
    // background class of a volume: 2.2 M candidates = 16 block shuffles of 138 000 elements, 2 ms that used to run inline with the GPU
    // idle; their generator consumption is value-independent too, case (b) of grid_draws)
    static const bool anchors_async = !(getenv("ARCO_SAMPLER_ANCHORS_ASYNC") && atoi(getenv("ARCO_SAMPLER_ANCHORS_ASYNC")) == 0);
    if ((shape >= 8192 || (anchors_async && high >= 65536)) && n_workers < max_threads && grid_draws(high, shape, cut, mirror, &draws, &kfix)) {
      copies[j] = g;
      MT* copy = &copies[j];
      int64_t* out = outs[j];
      const int par = shape >= 8192 ? 4 : 16;
      workers.run([copy, high, shape, cut, mirror, out, scratch, par, kfix]() { grid_sample_mt(*copy, high, shape, cut, mirror, out, scratch, par, kfix); });
      ++n_workers;
      g.skip(draws);
      if (trace) fprintf(stderr, "[sampler] job %d (worker, %ld) launched at %.3f ms\n", j, shape, ms());
      continue;
    }
    if (grid_sample_mt(g, high, shape, cut, mirror, outs[j], scratch, max_threads) == 0) { done = j; break; }
    if (trace) fprintf(stderr, "[sampler] job %d (inline, high %ld) done at %.3f ms\n", j, high, ms());
  }
  if (!deferred || done != n_jobs) { many_finish(); if (trace) fprintf(stderr, "[sampler] joined at %.3f ms\n", ms()); }
  if (done >= 0) store_state(state, g);
  return done;
}

// Pregenerate the generator's next state blocks for >= n_draws draws from `state` (not modified) - called right before
// the host blocks on the GPU's counters, so the ~1 ns per draw of mt19937 state regeneration (1.6 M draws per step at
// config 2) is paid while the host would be idle; background != 0: in a worker thread.  The next arco_grid_sample_many
// call that starts from exactly this state uses the blocks (any other state: they are ignored).  Returns the block count.
long arco_mt_pregen(const uint8_t* state, long state_bytes, long n_draws, int background) {
  MT base;
  if (n_draws <= 0 || !load_state(state, state_bytes, base)) return -1;
  { std::lock_guard<std::mutex> lock(g_arena_mutex); many_finish(); }     // (deferred workers may still read the old blocks)
  std::lock_guard<std::mutex> lk(g_stream_mutex);
  stream_join();
  const long blocks = n_draws / MT_N + 2;
  if ((long)g_stream.size() < blocks * MT_N) g_stream.resize((size_t)blocks * MT_N);
  g_stream_base = base; g_stream_base.stream = nullptr; g_stream_base.blk = -1;
  g_stream_blocks = blocks;
  uint32_t* dst = g_stream.data();
  auto work = [base, blocks, dst]() mutable {
    for (long b = 0; b < blocks; ++b) { base.next_state(); memcpy(dst + b * MT_N, base.st, sizeof(base.st)); }
  };
  if (background) g_stream_group.run(work); else work();
  return blocks;
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

// Advance the serialized torch CPU generator by n 32-bit draws without producing them (state regeneration only, ~0.5 us per
// 624 draws): the generator consumption of a tensor-library call whose VALUES this build does not need - the reference's
// `random_pool = torch.randn(K, 496, H, W)` (train_arco_2d.py:156) when the revisiting term is off - so that every later
// draw (weight initialisation, samplers, warps) sits where the reference's sits.
long arco_mt_skip(uint8_t* state, long state_bytes, uint64_t n) {
  MT g;
  if (!load_state(state, state_bytes, g)) return -1;
  g.skip(n);
  store_state(state, g);
  return 0;
}

}  // extern "C"
