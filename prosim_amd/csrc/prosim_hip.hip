// libprosim_hip.so -- host side of the MI355X closed-loop rollout engine + the C ABI
// declared in include/prosim_hip.h.  gfx950 only; build: see __graft_entry__.build().
#include "prosim_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <string>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "ps_attn.h"
#include "ps_kernels.h"
#include "ps_chain16.h"
#include "ps_pe_learn.h"
#include "ps_rowtile.h"

using namespace ps;

static thread_local std::string g_err;
// Experiment switches (ablations, forced kernel choices, in-kernel clocks) are read from the environment ONLY in tools builds
// (-DPS_EXPERIMENTS: tools/README.md).  The product library compiles them out -- a stray PS_* variable cannot change which kernel
// runs or corrupt a rollout -- and says so once, loudly, when it finds one set.
static const char* const kExpEnv[] = {"PS_C16_ABL", "PS_XCD", "PS_CHAIN_T", "PS_CHAIN_TP", "PS_CHAIN_T1", "PS_CHAIN_FLAGS", "PS_CHAIN_PROF",
                                      "PS_C16_ROWS", "PS_C16_ROWS_SMALL", "PS_C16_ROWS_S2S", "PS_S2S_C16", "PS_NO_SPLIT", "PS_SPLIT_MIN", "PS_SKIP_S2S_EDGE", "PS_POL_EDGE_PROBE"};
#ifdef PS_EXPERIMENTS
static const char* exp_env(const char* name) { return getenv(name); }
#else
static const char* exp_env(const char*) { return nullptr; }
#endif
static void warn_ignored_experiment_env() {
#ifndef PS_EXPERIMENTS
  static bool done = false;
  if (done) return;
  done = true;
  for (const char* n : kExpEnv)
    if (getenv(n)) fprintf(stderr, "[prosim_hip] %s is set, but this is the product build: experiment switches are compiled out and IGNORED "
                                   "(tools builds: hipcc -DPS_EXPERIMENTS, tools/README.md)\n", n);
#endif
}
extern "C" const char* ps_last_error(void) { return g_err.c_str(); }
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(x)                                                                                  \
  do {                                                                                             \
    hipError_t _e = (x);                                                                           \
    if (_e != hipSuccess)                                                                          \
      return fail(PS_E_HIP, std::string(#x) + ": " + hipGetErrorString(_e) + " @" + std::to_string(__LINE__)); \
  } while (0)

namespace {

thread_local hipStream_t g_free_sync = nullptr;   // see DevBuf::ensure

template <class T>
struct DevBuf {   // owning device allocation: freed by release() or at scope exit; movable, not copyable
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  int ensure(size_t count) {
    if (count <= n && p) return 0;
    // (round 5: a setter may run while the engine's previous rollout is still in flight -- a buffer that has to grow is freed only
    // after the stream has drained; StageScope names the stream)
    if (p && g_free_sync) (void)hipStreamSynchronize(g_free_sync);
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    size_t c = std::max<size_t>(count, 16);
    if (hipMalloc((void**)&p, c * sizeof(T)) != hipSuccess) return -1;
    n = c;
#ifdef PS_EXPERIMENTS   // (round 6: PS_POISON_ALLOC=<byte> fills every new device buffer with that byte -- a kernel that reads a buffer before anything
    // wrote it then finds 0x7f7f7f7f-class indices / 3e38-class floats instead of whatever the block held before: tools/gpu_r6_poison_alloc.sh)
    if (const char* pz = getenv("PS_POISON_ALLOC")) {   // (hipMemset runs on the null stream and does not wait: drain the device around it)
      (void)hipDeviceSynchronize();
      (void)hipMemset(p, (int)strtol(pz, nullptr, 0) & 0xff, c * sizeof(T));
      (void)hipDeviceSynchronize();
    }
#endif
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

struct EdgeSet {  // CSR by destination + normalised rel-PE
  DevBuf<int> cnt, eoff, toff, tdst, esrc, edst;   // toff: offsets in 32-edge tiles (sum of ceil(deg/32)); tdst: tile -> destination
  DevBuf<_Float16> rtA, rtT;                  // rel-PE rows (split fp16) as the two MFMA operand images (32-edge tiles)
  DevBuf<EdgeGeo> geo;                        // per-edge geometry records: what k_chain16 rebuilds the rel-PE rows from
  DevBuf<int> sync;                           // k_radius_geo: published counts per query (64 bits each) + the done counter (zero between launches)
  size_t cap_edges = 0;
  int nq = 0;
  int maxdeg = 0;
};

struct Stage { float ms = 0; };

// Upload staging of an engine (round 5: serving a stream of NEW batches).  hipMemcpyAsync from pageable memory stages and waits inside
// the call, and ~25 separate copies on the engine's stream are ~0.5 ms during which that stream runs no rollout.  Here a setter's
// uploads are copied into a PINNED host arena (the caller's arrays are free when the setter returns), cross PCIe as ONE copy into the
// arena's device mirror on a separate upload stream -- concurrently with the rollout the engine may still have in flight, whose
// inputs must not be touched yet --, and reach their destinations through one scatter kernel on the engine's stream behind that
// rollout.  No stream synchronisation ends the setter.  Two arenas per engine in turn (ps_set_scene switches): `done` marks the
// last scatter of an arena; begin() -- two ps_set_scene calls later -- waits for that mark; take(): a request that does not fit
// flushes, drains the stream, starts over and grows if it must.
struct UpSeg { unsigned char* dst; const unsigned char* src; unsigned long long bytes; };
constexpr int UP_SEGS = 24;
struct UpSegs { UpSeg s[UP_SEGS]; };
__global__ __launch_bounds__(256) void k_upload_scatter(UpSegs g) {   // blockIdx.y = segment; 16-byte pieces (arena slices and hipMalloc blocks are 256-byte aligned)
  const UpSeg sg = g.s[blockIdx.y];
  const size_t n16 = sg.bytes >> 4;
  const uint4* s4 = reinterpret_cast<const uint4*>(sg.src);
  uint4* d4 = reinterpret_cast<uint4*>(sg.dst);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d4[i] = s4[i];
  if (blockIdx.x == 0 && threadIdx.x < (sg.bytes & 15)) sg.dst[(n16 << 4) + threadIdx.x] = sg.src[(n16 << 4) + threadIdx.x];
}
// Device-to-device copies and zero fills INSIDE a rollout are kernels, not hipMemcpyAsync / hipMemsetAsync (round 5).  As graph nodes
// the runtime's copy / memset nodes proved fragile: an instantiated rollout graph replayed with the results of a rollout whose
// resets had not run (agent poses not back at their initial values, trajectory state not cleared: deterministic, and wrong) after
// an unrelated torch index_copy_ on the default stream of a process that had used RCCL -- while the same stages launched eagerly
// stayed right (bench.py PS_BENCH_FORCE_DIST with the per-rank diagnostic; tools/gpu_race_hunt.py).  A graph of kernel nodes only
// has nothing but in-queue ordering to rely on, and a 12-byte-per-agent copy is cheaper as a kernel than as an SDMA hop anyway.
__global__ __launch_bounds__(256) void k_dev_copy(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n16, size_t n4) {
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (n16) {
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (size_t i = i0; i < n16; i += stride) d4[i] = s4[i];
  } else {
    for (size_t i = i0; i < n4; i += stride) dst[i] = src[i];
  }
}
__global__ __launch_bounds__(256) void k_dev_zero(uint32_t* __restrict__ dst, size_t n16, size_t n4) {
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (n16) {
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    for (size_t i = i0; i < n16; i += stride) d4[i] = make_uint4(0u, 0u, 0u, 0u);
  } else {
    for (size_t i = i0; i < n4; i += stride) dst[i] = 0u;
  }
}
// bytes: a multiple of 4 (every caller moves floats, ints or pairs of halfs); 16-byte pieces when pointers and size allow
int dev_copy(hipStream_t st, void* dst, const void* src, size_t bytes) {
  if (!bytes) return 0;
  const bool v16 = ((uintptr_t)dst % 16 == 0) && ((uintptr_t)src % 16 == 0) && (bytes % 16 == 0);
  const size_t n = v16 ? bytes / 16 : bytes / 4;
  const unsigned grid = (unsigned)std::min<size_t>(512, (n + 255) / 256);
  hipLaunchKernelGGL(k_dev_copy, dim3(grid), dim3(256), 0, st, static_cast<uint32_t*>(dst), static_cast<const uint32_t*>(src), v16 ? n : 0, v16 ? 0 : n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int dev_zero(hipStream_t st, void* dst, size_t bytes) {
  if (!bytes) return 0;
  const bool v16 = ((uintptr_t)dst % 16 == 0) && (bytes % 16 == 0);
  const size_t n = v16 ? bytes / 16 : bytes / 4;
  const unsigned grid = (unsigned)std::min<size_t>(512, (n + 255) / 256);
  hipLaunchKernelGGL(k_dev_zero, dim3(grid), dim3(256), 0, st, static_cast<uint32_t*>(dst), v16 ? n : 0, v16 ? 0 : n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// two small segments in ONE launch (a launch is ~5 us of a single-scene rollout whatever it moves): copy, or zero fill where the source is null
struct CopySeg { uint32_t* dst; const uint32_t* src; size_t n4; };
__global__ __launch_bounds__(256) void k_dev_copy2(CopySeg a, CopySeg b) {
  const CopySeg& s = blockIdx.y ? b : a;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < s.n4; i += (size_t)gridDim.x * blockDim.x) s.dst[i] = s.src ? s.src[i] : 0u;
}
int dev_copy2(hipStream_t st, void* d0, const void* s0, size_t b0, void* d1, const void* s1, size_t b1) {
  const CopySeg a{static_cast<uint32_t*>(d0), static_cast<const uint32_t*>(s0), b0 / 4}, b{static_cast<uint32_t*>(d1), static_cast<const uint32_t*>(s1), b1 / 4};
  const size_t n = std::max(a.n4, b.n4);
  if (!n) return 0;
  hipLaunchKernelGGL(k_dev_copy2, dim3((unsigned)std::min<size_t>(256, (n + 255) / 256), 2), dim3(256), 0, st, a, b);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// one upload stream per device and process (an extra stream PER ENGINE costs the pipelined loop more than it hides: DESIGN.md section 7, round 3)
hipStream_t upload_stream(int device) {
  static std::mutex mu;
  static std::unordered_map<int, hipStream_t> streams;
  std::lock_guard<std::mutex> lock(mu);
  auto it = streams.find(device);
  if (it != streams.end()) return it->second;
  hipStream_t st = nullptr;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
  streams[device] = st;
  return st;
}
// Engine streams are pooled per device and live as long as the process: how the runtime spreads streams over the hardware queues
// depends on the ORDER in which a process created them, and four rollouts in flight on streams created after a few others had come
// and gone measured 20 M agent-steps/s instead of 26 M (tools/gpu_pipeline_depth.py with PS_DEPTHS=3,4 against 4, round 5).  With
// the pool -- and the upload stream always created first -- the k-th engine alive gets the k-th stream the process ever made.
struct EngineStreams {
  std::mutex mu;
  std::unordered_map<int, std::vector<std::pair<hipStream_t, bool>>> by_dev;   // (stream, in use)
};
EngineStreams& engine_streams() { static EngineStreams es; return es; }
hipStream_t engine_stream_acquire(int device) {
  (void)upload_stream(device);
  EngineStreams& es = engine_streams();
  std::lock_guard<std::mutex> lock(es.mu);
  auto& v = es.by_dev[device];
  for (auto& s : v)
    if (!s.second) { s.second = true; return s.first; }
  hipStream_t st = nullptr;
  // non-blocking: the legacy null stream (synchronous hipMemcpy, another library's default-stream work) does not
  // serialise against an engine's stream -- two engines on one GPU overlap; every read-back syncs explicitly
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
  v.push_back({st, true});
  return st;
}
void engine_stream_release(int device, hipStream_t st) {
  EngineStreams& es = engine_streams();
  std::lock_guard<std::mutex> lock(es.mu);
  for (auto& s : es.by_dev[device])
    if (s.first == st) s.second = false;
}
struct PinStage {
  unsigned char *p = nullptr, *dev = nullptr;   // the pinned host arena and its device mirror
  size_t cap = 0, used = 0, flushed = 0;
  hipEvent_t done = nullptr, h2d = nullptr;
  bool pending = false;
  int device = 0;
  std::vector<UpSeg> segs;   // slices taken since the last flush: (destination, OFFSET in the arena, bytes)
  int begin(hipStream_t st) {
    if (pending) { if (hipEventSynchronize(done) != hipSuccess) return -1; }
    else if (used) { if (hipStreamSynchronize(st) != hipSuccess) return -1; }   // (a setter that failed half-way left no mark)
    pending = false;
    used = flushed = 0;
    segs.clear();
    return 0;
  }
  int flush(hipStream_t st) {
    if (segs.empty()) return 0;
    hipStream_t up = upload_stream(device);
    if (!up) return -1;
    if (!h2d && hipEventCreateWithFlags(&h2d, hipEventDisableTiming) != hipSuccess) return -1;
    if (hipMemcpyAsync(dev + flushed, p + flushed, used - flushed, hipMemcpyHostToDevice, up) != hipSuccess) return -1;
    if (hipEventRecord(h2d, up) != hipSuccess || hipStreamWaitEvent(st, h2d, 0) != hipSuccess) return -1;
    for (size_t i0 = 0; i0 < segs.size(); i0 += UP_SEGS) {
      UpSegs g{};
      const int n = (int)std::min<size_t>(UP_SEGS, segs.size() - i0);
      for (int i = 0; i < n; ++i) g.s[i] = UpSeg{segs[i0 + i].dst, dev + (size_t)(uintptr_t)segs[i0 + i].src, segs[i0 + i].bytes};
      hipLaunchKernelGGL(k_upload_scatter, dim3(48, n), dim3(256), 0, st, g);
    }
    segs.clear();
    flushed = used;
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  void* take(size_t bytes, hipStream_t st) {
    const size_t need = (bytes + 255) & ~size_t(255);
    if (used + need > cap) {
      if (flush(st) || hipStreamSynchronize(st) != hipSuccess) return nullptr;   // every earlier copy out of this arena is complete
      pending = false;
      used = flushed = 0;
      if (need > cap) {
        if (p) (void)hipHostFree(p);
        if (dev) (void)hipFree(dev);
        p = dev = nullptr;
        cap = 0;
        const size_t want = std::max(need * 2, (size_t)1 << 24);
        if (hipHostMalloc((void**)&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return nullptr; }
        if (hipMalloc((void**)&dev, want) != hipSuccess) { (void)hipHostFree(p); p = dev = nullptr; return nullptr; }
        cap = want;
      }
    }
    void* r = p + used;
    used += need;
    return r;
  }
  int stage(void* dst, const void* h, size_t bytes, hipStream_t st) {
    unsigned char* pin = static_cast<unsigned char*>(take(bytes, st));
    if (!pin) return -1;
    std::memcpy(pin, h, bytes);
    segs.push_back(UpSeg{static_cast<unsigned char*>(dst), reinterpret_cast<const unsigned char*>((uintptr_t)(pin - p)), bytes});
    return 0;
  }
  int mark(hipStream_t st) {
    if (flush(st)) return -1;
    if (!done && hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return -1;
    if (hipEventRecord(done, st) != hipSuccess) return -1;
    pending = true;
    return 0;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    if (dev) (void)hipFree(dev);
    if (done) (void)hipEventDestroy(done);
    if (h2d) (void)hipEventDestroy(h2d);
    p = dev = nullptr; done = h2d = nullptr; cap = used = flushed = 0; pending = false;
    segs.clear();
  }
};
thread_local PinStage* g_stage = nullptr;   // the arena upload() stages through while a StageScope is alive (the setters of one engine)
// (round 6, ADVICE round 5: a setter that FAILS after it staged slices must not leave them in the arena -- the next successful mark() would
// scatter them to destinations that may have been re-allocated since.  A scope that is not commit()ted drops what was staged inside it; scopes
// nest on one arena: ps_set_drag_points opens one around its own uploads and rebuild_conditions' inner scope appends to the same arena.)
struct StageScope {
  PinStage* prev;
  hipStream_t prev_sync;
  PinStage* mine;
  size_t segs0, flushed0;
  bool ok = false;
  StageScope(PinStage* s, hipStream_t st) : prev(g_stage), prev_sync(g_free_sync), mine(s), segs0(s->segs.size()), flushed0(s->flushed) { g_stage = s; g_free_sync = st; }
  void commit() { ok = true; }
  ~StageScope() {
    if (!ok) {   // slices staged since entry and not yet flushed: forget them (a flush in between restarted the list: everything in it is ours)
      const size_t keep = mine->flushed == flushed0 ? std::min(segs0, mine->segs.size()) : 0;
      mine->segs.resize(keep);
    }
    g_stage = prev;
    g_free_sync = prev_sync;
  }
};

}  // namespace

struct ps_engine {
  ps_config cfg{};
  hipStream_t stream = nullptr;
  // ---- weights
  std::unordered_map<std::string, std::pair<const float*, int64_t>> src;  // only valid during create
  std::vector<float> arena_h;
  float* arena_d = nullptr;
  std::vector<size_t*> fixups;
  std::vector<AttnW> a2a, s2s, p2p, s2p, a2p, m2p, cnd;
  std::vector<AttnW> all_layers;
  AttnW* d_layers = nullptr;  // all layers, device copy (order: a2a s2s p2p s2p a2p m2p cnd)
  int L_a2a = 0, L_s2s = 0, L_p2p = 0, L_s2p = 0, L_a2p = 0, L_m2p = 0, L_cnd = 0;
  PointNetW pn_map{}, pn_obs{}, pn_drag{};   // pn_drag: DragPointEncoder (condition_encoders.py:152), optional
  Mlp3W mlp_prompt{}, mlp_pred{}, mlp_goal_prob{}, mlp_goal_point{};   // (goal heads: decoder/base.py:18-20, optional)
  HeadW head{};
  DevBuf<float> io_q, io_qt, io_cq, io_ar, io_av, io_l, io_s, io_g, io_m;   // split path / k_chain16: per-destination vectors (EdgeIO)
  CondW cond{};
  // learnable relative-PE embeddings (*.ATTN.LEARNABLE_PE): a2a, s2s | p2p, s2p | a2p, m2p; pe_on[part] = that part has them
  PeLearnW pe_learn[6]{};
  bool pe_on[3] = {false, false, false};
  Mlp3W mlp_obs_fuse{};                     // scene_encoder.obs_update_mlp (OBS_UPDATE.FUSION 'mlp')
  EdgeSet e_ua, e_um;                       // OBS_UPDATE.ATTN_UPDATE: agents <- agents (no self loops), agents <- map
  DevBuf<float> d_obs_new, d_kv_um;         // re-encoded observation rows; k|v of the map tokens for the s2s layers
  DevBuf<_Float16> d_kh_um;
  int step_upd = 0;
  const float* div32 = nullptr;
  // ---- scene
  bool have_scene = false, encoded = false, generated = false, reset = false;
  int B = 0, M = 0, P = 0, N = 0, Mv = 0, A = 0;
  // ps_set_replicas: the ONE input scene is rolled out `replicas` times side by side.  A = replicas * Ap agent rows
  // (replica-major), the Mv map tokens exist once and every replica's candidate range names them; encode_scene and
  // generate_policy run over the first Ap rows only (the replicas are identical until the first mode draw) and fan
  // their results out.  Without replicas Ap == A.
  int replicas = 1, replicas_next = 1, Ap = 0;
  std::vector<int> agent_slot;   // per agent row: its slot in the caller's per-agent arrays ([B or replicas][N])
  int maxA_scene = 0, maxM_scene = 0;
  std::vector<int> map_rows, agent_rows, agent_scene, map_scene, moff, aoff;  // host copies
  DevBuf<float> d_map_input, d_obs_input, d_prompt, d_fut;
  DevBuf<uint8_t> d_map_mask, d_obs_mask;
  DevBuf<int> d_map_rows, d_agent_rows, d_tok_scene, d_agent_type, d_r_map, d_r_agent, d_r_zero;
  DevBuf<float> d_tok, d_tok_pos, d_tok_ori, d_init_pos, d_init_head, d_cur_pos, d_cur_ori, d_prompt_pos, d_prompt_ori;
  DevBuf<float> d_xp, d_emd, d_xc, d_fused, d_obs_in, d_static_in, d_kv, d_kv_s2p, d_kv_m2p, d_kv_a2p;
  DevBuf<float> d_traj, d_vel, d_motion, d_reconst, d_goal_prob, d_goal_point, d_world;
  DevBuf<int> d_choice;                     // [R][A] motion mode each agent follows at each replan (ps_set_mode_choice; zeros = mode 0)
  DevBuf<float> d_noise;                    // [R][A][K][target_steps][2] action noise (ps_set_action_noise), used while have_noise
  bool have_noise = false;
  DevBuf<_Float16> d_kh, d_kh_s2p, d_kh_m2p, d_kh_a2p;   // split-fp16 k rows beside each kv buffer
  EdgeSet e_a2a, e_s2s, e_p2p, e_s2p, e_a2p, e_m2p, e_cnd;
  DevBuf<ChainStep> d_steps;
  std::vector<ChainStep> h_steps;
  int step_a2a = 0, step_s2s = 0, step_dec = 0, step_cnd = 0, step_pol = 0;
  // conditions
  bool have_cond = false;
  int n_cond_edges = 0;
  DevBuf<int> d_ent_off, d_ent_type;
  DevBuf<float> d_ent_val;
  // host copy of the condition entries (goal / tag from ps_set_conditions, drag from ps_set_drag_points); the device
  // CSR is rebuilt from both whenever either call changes them
  struct CondEnt { int agent, type, id; float v[3]; int src = -1; };   // src < 0: a unary condition (self loop on `agent`)
  std::vector<CondEnt> ents_gt, ents_drag, ents_pair;
  DevBuf<CondEdge> d_cond_edges;            // one record per edge of the condition graph (k_cond_edges)
  int n_cond_tiles = 0;
  // a condition TYPE that is present in the batch makes the reference run the condition layers over every policy
  // agent, even when each of its entries is masked off (condition_transformer/base.py:43-49, condition_attns.py:203-204)
  bool cond_present_gt = false, cond_present_drag = false, cond_present_pair = false;
  int n_drag = 0, drag_T = 0;
  DevBuf<float> d_drag_in, d_drag_emd;      // [n_drag][T][2] (NaN -> 0), [n_drag][128]
  DevBuf<uint8_t> d_drag_mask;              // [n_drag][T]
  bool have_fut = false;
  // log-replay agents: observed agents that are not policy agents (prompt_mask false on an observed slot).  Every
  // observed agent is a ROW of the per-agent buffers; is_policy marks the rows the simulation drives.
  int n_policy = 0;
  bool all_policy = true, have_log = false;
  std::vector<int> is_policy_h;
  DevBuf<int> d_is_policy, d_tok_live, d_live0;   // d_live0: agent row is in the scene at the initial step
  std::vector<uint8_t> declared_rows;             // ps_declare_agent_rows: consumed by the next ps_set_scene
  std::vector<int> live0_h;
  bool have_dead0 = false;
  DevBuf<uint8_t> d_obs_in_mask, d_fut_mask, d_obs_mask_rows;   // (d_obs_mask_rows: the initial mask, one row per agent)
  DevBuf<float> d_fut_pos, d_fut_head;
  int stride_steps = 0;
  // timing of the dominant kernel
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool time_chain = false;
  double chain_ms_sum = 0;
  int chain_launches = 0;
  float edge_counts[8] = {0};
  // whole-rollout hipGraph (captured on the first ps_rollout after a scene / condition change)
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  int force_mt = 0;           // ps_test_pointnet_mt (test hook): row tiles per wave of the row-tile PointNet, -1 = the staged kernel
  int node_mt = 0;            // ps_set_row_impl(10 + mt): row tiles per wave of the row-tile node kernels forced to mt (experiments, tests)
  bool wg_edges = false;      // ps_set_row_impl(2): the split path's edge half on the 16-row workgroup kernel (k_edge16) instead of k_edge_rows (A/B, cross-check)
  int env_ready = -1;         // the replan whose step_env already ran in the previous replan's head launch (k_policy_head_row's tail), or -1
  int search_impl = 0;        // ps_set_search_impl: 0 = a radius search with geometry records is ONE launch (k_radius_geo), 1 = count / fill / record launches (rounds 1-4; A/B, cross-check), 2 = as 0, the look-back recomputes instead of waiting (tests)
  bool legacy_rows = false;   // ps_set_row_impl(1): the round-3 staged row kernels (k_pointnet_mfma, k_node) instead of the row-tile ones (A/B and parity tools)
  int chain_rows = 0;   // ps_set_chain_rows: 0 = latency-optimal choice, else rows per workgroup of the fused attention launches
  int chain_impl = 0;   // ps_set_chain_impl: 0 = by mode (k_chain16 in throughput mode: chain_rows >= 8; k_attn_chain otherwise), 1 = k_attn_chain, 2 = k_chain16, 3 = k_chain16 + the encoder's s2s layers on it
  bool graph_ok = false;
  bool use_graph = true;
  // ps_enable_policy_events: an event pair around every policy-chain launch of a rollout, recorded on the engine's stream
  // (the rollout is then launched eagerly), so the launch durations of a PIPELINED run can be read afterwards
  bool policy_events = false;
  std::vector<hipEvent_t> pev;
  // round 5: serving a stream of new batches -- pinned upload staging, the captured graph kept across scenes of one shape, results
  // copied back behind the rollout
  PinStage stage[2];                                  // two arenas in turn (ps_set_scene switches): a batch can be uploaded behind a rollout still in flight
  int stage_cur = 0;
  std::vector<uint64_t> graph_sig;                    // what the captured launch sequence depends on (rollout_signature)
  int64_t graph_captures = 0, graph_reuses = 0;       // (ps_graph_stats: how often ps_rollout had to capture / could keep the graph)
};

namespace {
void drop_graph(ps_engine* e);      // the captured rollout must be re-checked against rollout_signature() before its next replay
void destroy_graph(ps_engine* e);
int io_for(ps_engine* e, int Nd, EdgeIO& io);
// the learnable rel-PE embedding that makes the rows of edge set `es`, or nullptr (fixed Fourier rows)
const PeLearnW* pe_of(const ps_engine* e, const EdgeSet* es) {
  const EdgeSet* sets[6] = {&e->e_a2a, &e->e_s2s, &e->e_p2p, &e->e_s2p, &e->e_a2p, &e->e_m2p};
  for (int i = 0; i < 6; ++i)
    if (es == sets[i]) return e->pe_on[i / 2] ? &e->pe_learn[i] : nullptr;
  // ATTN_UPDATE's sets take the scene encoder's embeddings: agents <- agents a2a_rel_pe_emb, agents <- map s2s_rel_pe_emb
  // (attn_fusion.py:158-159 through :44-76)
  if (es == &e->e_ua) return e->pe_on[0] ? &e->pe_learn[0] : nullptr;
  if (es == &e->e_um) return e->pe_on[0] ? &e->pe_learn[1] : nullptr;
  return nullptr;
}
}
// ------------------------------------------------------------------------------------------ weights
namespace {

struct Builder {
  ps_engine* e;
  std::string err;
  const float* get(const std::string& name, int64_t numel) {
    auto it = e->src.find(name);
    if (it == e->src.end()) {
      if (err.empty()) err = "missing weight '" + name + "'";
      return nullptr;
    }
    if (it->second.second != numel) {
      if (err.empty())
        err = "weight '" + name + "' has " + std::to_string(it->second.second) + " elements, expected " + std::to_string(numel);
      return nullptr;
    }
    return it->second.first;
  }
  bool has(const std::string& name) { return e->src.count(name) != 0; }
  // append to the host arena; returns offset (floats), 64-float aligned
  size_t put(const std::vector<float>& v) {
    size_t off = (e->arena_h.size() + 63) & ~size_t(63);
    e->arena_h.resize(off + v.size());
    std::copy(v.begin(), v.end(), e->arena_h.begin() + off);
    return off;
  }
  // record a pointer slot that must be rebased onto the device arena
  void slot(const float** p, size_t off) {
    *p = reinterpret_cast<const float*>(off + 1);  // tag: offset+1 (0 stays nullptr)
    ptrs.push_back(p);
  }
  std::vector<const float**> ptrs;
  void plain(const float** p, const std::string& name, int64_t numel) {
    const float* s = get(name, numel);
    if (!s) return;
    slot(p, put(std::vector<float>(s, s + numel)));
  }
  // torch Linear weight [out = 128][in] -> split-fp16 MFMA B fragments over input columns [in0, in0+in_n), K padded
  // to a multiple of 32: [n-tile 8][k-block][hi|lo][lane 64][8], lane = (n & 15) + 16*kq holds k = 32*ks + 8*kq ..+8
  // perm: the K index inside a 32-wide k-block runs in the order the row-tile kernels hand a result tile on (ps_rowtile.h):
  // lane group kq, element j <-> k = 32 ks + 16 (j >> 2) + 4 kq + (j & 3) instead of 32 ks + 8 kq + j
  // (every fragment's lo half is scaled by 2^11, ps_device.h f16_los); perm: the permuted K order of the row-tile kernels
  void fragments(const _Float16** p, const std::string& name, int out, int in, int in0, int in_n, bool perm = false, bool rt = false) {
    const float* s = get(name, (int64_t)out * in);
    if (s) fragments_raw(p, s, out, in, in0, in_n, perm, rt);
  }
  void fragments_raw(const _Float16** p, const float* s, int out, int in, int in0, int in_n, bool perm = false, bool /*rt*/ = false) {
    const int k32 = (in_n + 31) / 32, nt_n = out / 16;
    std::vector<float> packed(((size_t)nt_n * k32 * 2 * 512 + 1) / 2);
    _Float16* h = reinterpret_cast<_Float16*>(packed.data());
    for (int nt = 0; nt < nt_n; ++nt)
      for (int ks = 0; ks < k32; ++ks)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int n = nt * 16 + (lane & 15), kq = lane >> 4;
            const int k = perm ? ks * 32 + 16 * (j >> 2) + 4 * kq + (j & 3) : ks * 32 + kq * 8 + j;
            const float v = k < in_n ? s[(size_t)n * in + in0 + k] : 0.f;
            const _Float16 hi = (_Float16)v;
            const size_t o = ((size_t)(nt * k32 + ks) * 2) * 512 + (size_t)lane * 8 + j;
            h[o] = hi;
            h[o + 512] = (_Float16)((v - (float)hi) * 2048.f);   // (ps_device.h f16_los: every GEMM fragment carries the scaled lo half)
          }
    slot(reinterpret_cast<const float**>(p), put(packed));
  }
  void transposed(const float** p, const std::string& name, int out, int in, int in0 = 0, int in_n = -1) {
    // torch Linear weight [out][in] -> K-major [in_n][out] over input columns [in0, in0+in_n)
    const float* s = get(name, (int64_t)out * in);
    if (!s) return;
    if (in_n < 0) in_n = in;
    std::vector<float> t((size_t)in_n * out);
    for (int k = 0; k < in_n; ++k)
      for (int n = 0; n < out; ++n) t[(size_t)k * out + n] = s[(size_t)n * in + in0 + k];
    slot(p, put(t));
  }
};

void build_attn(Builder& b, const std::string& p, AttnW& w) {
  std::memset(&w, 0, sizeof(w));
  b.plain(&w.ln_src_w, p + ".attn_prenorm_x_src.weight", D);
  b.plain(&w.ln_src_b, p + ".attn_prenorm_x_src.bias", D);
  const std::string dn = b.has(p + ".attn_prenorm_x_dst.weight") ? ".attn_prenorm_x_dst" : ".attn_prenorm_x_src";
  b.plain(&w.ln_dst_w, p + dn + ".weight", D);
  b.plain(&w.ln_dst_b, p + dn + ".bias", D);
  b.transposed(&w.Wq_t, p + ".to_q.weight", D, D);
  b.plain(&w.bq, p + ".to_q.bias", D);
  b.transposed(&w.Ws_t, p + ".to_s.weight", D, D);
  b.plain(&w.bs, p + ".to_s.bias", D);
  b.transposed(&w.Wgx_t, p + ".to_g.weight", D, 2 * D, D, D);
  b.transposed(&w.Wga_t, p + ".to_g.weight", D, 2 * D, 0, D);
  b.plain(&w.bg, p + ".to_g.bias", D);
  b.transposed(&w.Wout_t, p + ".to_out.weight", D, D);
  b.plain(&w.bout, p + ".to_out.bias", D);
  b.plain(&w.ln_post_w, p + ".attn_postnorm.weight", D);
  b.plain(&w.ln_post_b, p + ".attn_postnorm.bias", D);
  b.plain(&w.ln_ffpre_w, p + ".ff_prenorm.weight", D);
  b.plain(&w.ln_ffpre_b, p + ".ff_prenorm.bias", D);
  b.plain(&w.ln_ffpost_w, p + ".ff_postnorm.weight", D);
  b.plain(&w.ln_ffpost_b, p + ".ff_postnorm.bias", D);
  b.transposed(&w.W1_t, p + ".ff_mlp.0.weight", FF, D);
  b.plain(&w.b1, p + ".ff_mlp.0.bias", FF);
  b.transposed(&w.W2_t, p + ".ff_mlp.3.weight", D, FF);
  b.plain(&w.b2, p + ".ff_mlp.3.bias", D);
  // folds of the relative-PE LayerNorm affine into to_k_r / to_v_r
  const float* g = b.get(p + ".attn_prenorm_r.weight", D);
  const float* be = b.get(p + ".attn_prenorm_r.bias", D);
  const float* wkr = b.get(p + ".to_k_r.weight", (int64_t)D * D);
  const float* wvr = b.get(p + ".to_v_r.weight", (int64_t)D * D);
  const float* bvr = b.get(p + ".to_v_r.bias", D);
  const float* wk = b.get(p + ".to_k.weight", (int64_t)D * D);
  const float* wv = b.get(p + ".to_v.weight", (int64_t)D * D);
  const float* bv = b.get(p + ".to_v.bias", D);
  if (!g || !be || !wkr || !wvr || !bvr || !wk || !wv || !bv) return;
  std::vector<float> wkrg((size_t)D * D), kb(D), wvrgt((size_t)D * D), vb(D), wkv((size_t)D * 2 * D), bkv(2 * D, 0.f);
  for (int hd = 0; hd < D; ++hd) {
    double sk = 0, sv = 0;
    for (int c = 0; c < D; ++c) {
      wkrg[(size_t)hd * D + c] = wkr[(size_t)hd * D + c] * g[c];
      wvrgt[(size_t)c * D + hd] = wvr[(size_t)hd * D + c] * g[c];
      sk += (double)wkr[(size_t)hd * D + c] * be[c];
      sv += (double)wvr[(size_t)hd * D + c] * be[c];
    }
    kb[hd] = (float)sk;
    vb[hd] = (float)(sv + bvr[hd]);
    bkv[D + hd] = bv[hd];
    for (int k = 0; k < D; ++k) {
      wkv[(size_t)k * 2 * D + hd] = wk[(size_t)hd * D + k];
      wkv[(size_t)k * 2 * D + D + hd] = wv[(size_t)hd * D + k];
    }
  }
  b.slot(&w.Wkr_g, b.put(wkrg));
  b.slot(&w.kb, b.put(kb));
  b.slot(&w.Wvr_gt, b.put(wvrgt));
  {   // folded variants for geometric rel-PE rows (features 96..127 == 64..95)
    std::vector<float> wkrg3 = wkrg, wvrgt3 = wvrgt;
    for (int hd = 0; hd < D; ++hd)
      for (int i = 0; i < 32; ++i) {
        wkrg3[(size_t)hd * D + 64 + i] += wkrg3[(size_t)hd * D + 96 + i];
        wvrgt3[(size_t)(64 + i) * D + hd] += wvrgt3[(size_t)(96 + i) * D + hd];
      }
    b.slot(&w.Wkr_g3, b.put(wkrg3));
    b.slot(&w.Wvr_gt3, b.put(wvrgt3));
  }
  b.slot(&w.vb, b.put(vb));
  b.slot(&w.Wkv_t, b.put(wkv));
  {   // split path (k_node): node Linears as B fragments
    const float* wq = b.get(p + ".to_q.weight", (int64_t)D * D);
    const float* ws = b.get(p + ".to_s.weight", (int64_t)D * D);
    const float* wg = b.get(p + ".to_g.weight", (int64_t)D * 2 * D);
    if (wq && ws && wg) {
      std::vector<float> qsg((size_t)3 * D * D);
      std::copy(wq, wq + (size_t)D * D, qsg.begin());
      std::copy(ws, ws + (size_t)D * D, qsg.begin() + (size_t)D * D);
      for (int n = 0; n < D; ++n) std::copy(wg + (size_t)n * 2 * D + D, wg + (size_t)n * 2 * D + 2 * D, qsg.begin() + (size_t)(2 * D + n) * D);
      b.fragments_raw(&w.Fqsg, qsg.data(), 3 * D, D, 0, D);
    }
    b.fragments(&w.Fga, p + ".to_g.weight", D, 2 * D, 0, D);
    b.fragments(&w.Fout, p + ".to_out.weight", D, D, 0, D);
    b.fragments(&w.F1, p + ".ff_mlp.0.weight", FF, D, 0, D);
    b.fragments(&w.F2, p + ".ff_mlp.3.weight", D, FF, 0, FF);
    // q~: per head h one k-block = the 32 q columns (heads 2*(h/2), 2*(h/2)+1) with the other head's rows zero;
    // B[k][n] = Wkr_g[16h + d][c], k = 16*(h & 1) + d, n-tile nt -> c = 16 nt + n
    auto kr_frag = [&](const _Float16** dst, const std::vector<float>& wk_, int ntq) {
      std::vector<float> packed(((size_t)8 * ntq * 1024 + 1) / 2);
      _Float16* hh = reinterpret_cast<_Float16*>(packed.data());
      for (int h = 0; h < 8; ++h)
        for (int nt = 0; nt < ntq; ++nt)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int c = nt * 16 + (lane & 15), kl = (lane >> 4) * 8 + j;
              const float v = (kl >> 4) == (h & 1) ? wk_[(size_t)(h * 16 + (kl & 15)) * D + c] : 0.f;
              const _Float16 hi = (_Float16)v;
              const size_t o = (size_t)(h * ntq + nt) * 1024 + (size_t)lane * 8 + j;
              hh[o] = hi;
              hh[o + 512] = (_Float16)((v - (float)hi) * 2048.f);
            }
      b.slot(reinterpret_cast<const float**>(dst), b.put(packed));
    };
    // to_v_r fold: per head h the n-tile of columns 16h..16h+15 of Wvr_gt [c][hd]; B[k = c][n = d]
    auto vr_frag = [&](const _Float16** dst, const std::vector<float>& wv_, int kr, float lo_scale = 2048.f) {
      std::vector<float> packed(((size_t)8 * kr * 1024 + 1) / 2);
      _Float16* hh = reinterpret_cast<_Float16*>(packed.data());
      for (int h = 0; h < 8; ++h)
        for (int ks = 0; ks < kr; ++ks)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int c = ks * 32 + (lane >> 4) * 8 + j, d = lane & 15;
              const float v = wv_[(size_t)c * D + h * 16 + d];
              const _Float16 hi = (_Float16)v;
              const size_t o = (size_t)(h * kr + ks) * 1024 + (size_t)lane * 8 + j;
              hh[o] = hi;
              hh[o + 512] = (_Float16)((v - (float)hi) * lo_scale);
            }
      b.slot(reinterpret_cast<const float**>(dst), b.put(packed));
    };
    std::vector<float> wkrg3 = wkrg, wvrgt3 = wvrgt;
    for (int hd = 0; hd < D; ++hd)
      for (int i = 0; i < 32; ++i) {
        wkrg3[(size_t)hd * D + 64 + i] += wkrg3[(size_t)hd * D + 96 + i];
        wvrgt3[(size_t)(64 + i) * D + hd] += wvrgt3[(size_t)(96 + i) * D + hd];
      }
    {   // row-tile q~ (ps_rowtile.h): K = 16 fragments, A[m = c][k = d] = Wkr_g3[16 h + d][16 ct + m]; lane = m + 16 kq holds d = 4 kq .. 4 kq + 3
      std::vector<float> packed(((size_t)8 * 6 * 2 * 256 + 1) / 2);
      _Float16* hh = reinterpret_cast<_Float16*>(packed.data());
      for (int h = 0; h < 8; ++h)
        for (int ct = 0; ct < 6; ++ct)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
              const int c = ct * 16 + (lane & 15), d = (lane >> 4) * 4 + j;
              const float v = wkrg3[(size_t)(h * 16 + d) * D + c];
              const _Float16 hi = (_Float16)v;
              const size_t o = (size_t)(h * 6 + ct) * 512 + (size_t)lane * 4 + j;
              hh[o] = hi;
              hh[o + 256] = (_Float16)((v - (float)hi) * 2048.f);
            }
      b.slot(reinterpret_cast<const float**>(&w.Fkr3_x), b.put(packed));
    }
    if (wq && ws && wg) {
      std::vector<float> qsg((size_t)3 * D * D);
      std::copy(wq, wq + (size_t)D * D, qsg.begin());
      std::copy(ws, ws + (size_t)D * D, qsg.begin() + (size_t)D * D);
      for (int n = 0; n < D; ++n) std::copy(wg + (size_t)n * 2 * D + D, wg + (size_t)n * 2 * D + 2 * D, qsg.begin() + (size_t)(2 * D + n) * D);
      b.fragments_raw(&w.Fqsg_Q, qsg.data(), 3 * D, D, 0, D, true);
    }
    b.fragments(&w.Fga_Q, p + ".to_g.weight", D, 2 * D, 0, D, true);
    b.fragments(&w.Fout_Q, p + ".to_out.weight", D, D, 0, D, true);
    b.fragments(&w.F1_Q, p + ".ff_mlp.0.weight", FF, D, 0, D, true);
    b.fragments(&w.F2_Q, p + ".ff_mlp.3.weight", D, FF, 0, FF, true);
    kr_frag(&w.Fkr, wkrg, 8);
    kr_frag(&w.Fkr3, wkrg3, 6);
    vr_frag(&w.Fvr, wvrgt, 4);
    vr_frag(&w.Fvr3, wvrgt3, 3);
    vr_frag(&w.Fvr3_Q, wvrgt3, 3, 2048.f);
  }
  {   // [to_k ; to_v] as one [256][128] Linear -> B fragments
    std::vector<float> kvw((size_t)2 * D * D);
    std::copy(wk, wk + (size_t)D * D, kvw.begin());
    std::copy(wv, wv + (size_t)D * D, kvw.begin() + (size_t)D * D);
    b.fragments_raw(&w.Wkv_F, kvw.data(), 2 * D, D, 0, D);
    b.fragments_raw(&w.Wkv_Q, kvw.data(), 2 * D, D, 0, D, true);
  }
  b.slot(&w.bkv, b.put(bkv));
  // packed small vectors (ps_attn.h SP_* offsets)
  {
    std::vector<float> sp(SP_SIZE, 0.f);
    auto cp = [&](int off, const std::string& name, int n) {
      const float* s_ = b.get(name, n);
      if (s_) std::copy(s_, s_ + n, sp.begin() + off);
    };
    cp(SP_LN_DST_W, p + dn + ".weight", D);
    cp(SP_LN_DST_B, p + dn + ".bias", D);
    cp(SP_BQ, p + ".to_q.bias", D);
    cp(SP_BS, p + ".to_s.bias", D);
    cp(SP_BG, p + ".to_g.bias", D);
    std::copy(kb.begin(), kb.end(), sp.begin() + SP_KB);
    std::copy(vb.begin(), vb.end(), sp.begin() + SP_VB);
    cp(SP_BOUT, p + ".to_out.bias", D);
    cp(SP_LN_POST_W, p + ".attn_postnorm.weight", D);
    cp(SP_LN_POST_B, p + ".attn_postnorm.bias", D);
    cp(SP_LN_FFPRE_W, p + ".ff_prenorm.weight", D);
    cp(SP_LN_FFPRE_B, p + ".ff_prenorm.bias", D);
    cp(SP_B1, p + ".ff_mlp.0.bias", FF);
    cp(SP_B2, p + ".ff_mlp.3.bias", D);
    cp(SP_LN_FFPOST_W, p + ".ff_postnorm.weight", D);
    cp(SP_LN_FFPOST_B, p + ".ff_postnorm.bias", D);
    b.slot(&w.sp, b.put(sp));
  }
}

// sequential indices of the reference MLP's nn.Sequential (models/layers/mlp.py:475-494)
std::vector<std::pair<int, int>> mlp_seq(int n_lin, bool without_norm) {
  std::vector<std::pair<int, int>> r;
  int idx = 0;
  for (int i = 0; i < n_lin; ++i) {
    int lin = idx++, ln = -1;
    if (i < n_lin - 1) {
      if (!without_norm) ln = idx++;
      idx++;  // ReLU
    }
    r.push_back({lin, ln});
  }
  return r;
}

void build_pointnet(Builder& b, const std::string& p, int in_dim, int n_pre, int n_mlp, PointNetW& w) {
  std::memset(&w, 0, sizeof(w));
  w.in_dim = in_dim;
  w.n_pre = n_pre;
  w.n_mid = n_mlp - n_pre;
  auto sq = mlp_seq(n_pre, false);
  for (int l = 0; l < n_pre; ++l) {
    const int K = l == 0 ? in_dim : D;
    const std::string q = p + ".pre_mlps.mlp." + std::to_string(sq[l].first);
    b.transposed(&w.pre_Wt[l], q + ".weight", D, K);
    b.fragments(&w.pre_F[l], q + ".weight", D, K, 0, K);
    b.fragments(&w.pre_Q[l], q + ".weight", D, K, 0, K, l > 0, true);   // (layer 0 reads the input rows in the natural K order)
    b.plain(&w.pre_b[l], q + ".bias", D);
    if (sq[l].second >= 0) {
      const std::string n = p + ".pre_mlps.mlp." + std::to_string(sq[l].second);
      b.plain(&w.pre_lnw[l], n + ".weight", D);
      b.plain(&w.pre_lnb[l], n + ".bias", D);
    }
  }
  sq = mlp_seq(w.n_mid, false);
  for (int l = 0; l < w.n_mid; ++l) {
    const int K = l == 0 ? 2 * D : D;
    const std::string q = p + ".mlps.mlp." + std::to_string(sq[l].first);
    b.transposed(&w.mid_Wt[l], q + ".weight", D, K);
    b.fragments(&w.mid_F[l], q + ".weight", D, K, 0, D);   // layer 0: the point-feature half ...
    if (l == 0) b.fragments(&w.mid_P, q + ".weight", D, K, D, D);   // ... and the pooled half
    b.fragments(&w.mid_Q[l], q + ".weight", D, K, 0, D, true);
    if (l == 0) b.fragments(&w.mid_PQ, q + ".weight", D, K, D, D, true);
    b.plain(&w.mid_b[l], q + ".bias", D);
    if (sq[l].second >= 0) {
      const std::string n = p + ".mlps.mlp." + std::to_string(sq[l].second);
      b.plain(&w.mid_lnw[l], n + ".weight", D);
      b.plain(&w.mid_lnb[l], n + ".bias", D);
    }
  }
  b.fragments(&w.out_F0, p + ".out_mlps.mlp.0.weight", D, D, 0, D);
  b.fragments(&w.out_F1, p + ".out_mlps.mlp.2.weight", D, D, 0, D);
  b.fragments(&w.out_Q0, p + ".out_mlps.mlp.0.weight", D, D, 0, D, true);
  b.fragments(&w.out_Q1, p + ".out_mlps.mlp.2.weight", D, D, 0, D, true);
  b.transposed(&w.out_W0t, p + ".out_mlps.mlp.0.weight", D, D);
  b.plain(&w.out_b0, p + ".out_mlps.mlp.0.bias", D);
  b.transposed(&w.out_W1t, p + ".out_mlps.mlp.2.weight", D, D);
  b.plain(&w.out_b1, p + ".out_mlps.mlp.2.bias", D);
}

// FourierEmbedding(input_dim = 3, hidden_dim = 128, num_freq_bands = 64) (layers/fourier_embedding.py:11-35)
void build_pe_learn(Builder& b, const std::string& p, PeLearnW& w) {
  std::memset(&w, 0, sizeof(w));
  b.plain(&w.freqs, p + ".freqs.weight", 3 * 64);
  std::vector<float> b2(D, 0.f);
  for (int i = 0; i < 3; ++i) {
    const std::string q = p + ".mlps." + std::to_string(i);
    b.fragments(&w.F1[i], q + ".0.weight", D, 129, 0, 128);
    b.transposed(&w.w1x[i], q + ".0.weight", D, 129, 128, 1);
    b.plain(&w.b1[i], q + ".0.bias", D);
    b.plain(&w.ln1w[i], q + ".1.weight", D);
    b.plain(&w.ln1b[i], q + ".1.bias", D);
    b.fragments(&w.F2[i], q + ".3.weight", D, D, 0, D);
    const float* s = b.get(q + ".3.bias", D);
    if (s)
      for (int c = 0; c < D; ++c) b2[c] += s[c];
  }
  b.slot(&w.b2sum, b.put(b2));
  b.plain(&w.lnow, p + ".to_out.0.weight", D);
  b.plain(&w.lnob, p + ".to_out.0.bias", D);
  b.fragments(&w.Fo, p + ".to_out.2.weight", D, D, 0, D);
  b.plain(&w.bo, p + ".to_out.2.bias", D);
}

void build_mlp3(Builder& b, const std::string& p, std::vector<int> dims, bool without_norm, Mlp3W& m) {
  std::memset(&m, 0, sizeof(m));
  m.n = (int)dims.size() - 1;
  for (size_t i = 0; i < dims.size(); ++i) m.dims[i] = dims[i];
  auto sq = mlp_seq(m.n, without_norm);
  for (int l = 0; l < m.n; ++l) {
    const std::string q = p + ".mlp." + std::to_string(sq[l].first);
    b.plain(&m.W[l], q + ".weight", (int64_t)dims[l + 1] * dims[l]);
    b.plain(&m.b[l], q + ".bias", dims[l + 1]);
    if (sq[l].second >= 0) {
      const std::string n = p + ".mlp." + std::to_string(sq[l].second);
      b.plain(&m.lnw[l], n + ".weight", dims[l + 1]);
      b.plain(&m.lnb[l], n + ".bias", dims[l + 1]);
    }
  }
}

const char* kTagNames[11] = {"Stopping", "Accelerate", "Decelerate", "KeepSpeed", "LeftLaneChange", "RightLaneChange",
                             "KeepLane", "LeftTurn", "RightTurn", "Straight", "Parked"};

}  // namespace

extern "C" int ps_create(const ps_config* cfg, int32_t n_tensors, const char* const* names, const float* const* data,
                         const int64_t* numel, ps_engine** out) {
  if (!cfg || !out) return fail(PS_E_ARG, "null argument");
  warn_ignored_experiment_env();
  if (cfg->hidden != D || cfg->heads != H || cfg->head_dim != DH)
    return fail(PS_E_ARG, "this build supports hidden=128, heads=8, head_dim=16 only");
  if (cfg->hist_steps > 15 || cfg->obs_dim > 24 || cfg->map_dim > 24 || cfg->motion_k < 1 || cfg->motion_k > 16 || cfg->state_dim != 3 + (cfg->no_pred_vel ? 0 : 2) + (cfg->pred_gmm ? 3 : 0) ||
      cfg->target_steps * cfg->state_dim * (cfg->k_pred_mlp ? cfg->motion_k : 1) > 128 || cfg->map_pre_layers > 4 || cfg->obs_pre_layers > 4 ||
      cfg->map_mlp_layers - cfg->map_pre_layers > 4 || cfg->obs_mlp_layers - cfg->obs_pre_layers > 4)
    return fail(PS_E_ARG, "unsupported config (hist<=15, obs_dim<=24, map_dim<=24, 1<=motion_k<=16, steps*state (x motion_k with PRED_MODE mlp) <=128)");
  if (cfg->map_encoder_mlp || cfg->obs_encoder_mlp)
    return fail(PS_E_ARG, std::string("MODEL.SCENE_ENCODER.") + (cfg->map_encoder_mlp ? "MAP_TYPE" : "OBS_TYPE") +
                          " 'mlp' (scene_encoder/map_encoder.py:5, obs_encoder.py:19) is not built: this engine runs the 'pointnet' encoders only");
  if (cfg->goal_pred_k < 0 || cfg->goal_pred_k > 64) return fail(PS_E_ARG, "goal_pred_k must be in 0..64");
  if (cfg->no_pred_vel && cfg->replan_freq < 2)
    return fail(PS_E_ARG, "PRED_VEL False needs replan_freq >= 2 (velocities from position differences over hist_steps + 2 steps)");
  if (cfg->replan_freq < 1 || cfg->replan_freq > cfg->target_steps)
    return fail(PS_E_ARG, "replan_freq must be in 1..target_steps (a replan appends replan_freq of the target_steps predicted states)");
  if (cfg->pol_max_neigh < 1 || cfg->pol_max_neigh > 2047 || cfg->dec_max_neigh < 1 || cfg->dec_max_neigh > 2047)
    return fail(PS_E_ARG, "pol_max_neigh / dec_max_neigh must be in 1..2047 (64 rel-PE tiles per destination)");
  const bool any_lpe = cfg->enc_learnable_pe || cfg->dec_learnable_pe || cfg->pol_learnable_pe;
  if (any_lpe && cfg->pe_num_freq != 64)
    return fail(PS_E_ARG, "learnable rel-PE: this build supports PE_NUM_FREQ = 64 (the reference's default) only");
  if (hipSetDevice(cfg->device) != hipSuccess) return fail(PS_E_HIP, "hipSetDevice failed (no GPU?)");
  ps_engine* e = new ps_engine();
  e->cfg = *cfg;
  e->pe_on[0] = cfg->enc_learnable_pe != 0; e->pe_on[1] = cfg->dec_learnable_pe != 0; e->pe_on[2] = cfg->pol_learnable_pe != 0;
  for (int i = 0; i < n_tensors; ++i) e->src[names[i]] = {data[i], numel[i]};
  Builder b{e};
  auto layers = [&](const std::string& pre, int n, std::vector<AttnW>& v) {
    v.resize(n);
    for (int i = 0; i < n; ++i) build_attn(b, pre + "." + std::to_string(i), v[i]);
  };
  layers("scene_encoder.a2a_attn_layers", cfg->scene_layers, e->a2a);
  layers("scene_encoder.s2s_attn_layers", cfg->scene_layers, e->s2s);
  layers("decoder.p2p_attn_layers", cfg->dec_layers, e->p2p);
  layers("decoder.s2p_attn_layers", cfg->dec_layers, e->s2p);
  layers("policy.act_decoder.a2p_attn_layers", cfg->pol_layers, e->a2p);
  layers("policy.act_decoder.m2p_attn_layers", cfg->pol_layers, e->m2p);
  layers("condition_transformers.policy_decoder.condition_attn.attn_layers", cfg->cond_layers, e->cnd);
  build_pointnet(b, "scene_encoder.map_encoder", cfg->map_dim, cfg->map_pre_layers, cfg->map_mlp_layers, e->pn_map);
  build_pointnet(b, "scene_encoder.obs_encoder", cfg->obs_dim, cfg->obs_pre_layers, cfg->obs_mlp_layers, e->pn_obs);
  if (cfg->drag_mlp_layers > 0)
    build_pointnet(b, "condition_transformers.policy_decoder.condition_encoders.drag_point.pointnet_encoder", 2,
                   cfg->drag_pre_layers, cfg->drag_mlp_layers, e->pn_drag);
  {
    static const char* kPe[6] = {"scene_encoder.a2a_rel_pe_emb", "scene_encoder.s2s_rel_pe_emb", "decoder.p2p_rel_pe_emb",
                                 "decoder.s2p_rel_pe_emb", "policy.act_decoder.a2p_rel_pe_emb", "policy.act_decoder.m2p_rel_pe_emb"};
    for (int i = 0; i < 6; ++i)
      if (e->pe_on[i / 2]) build_pe_learn(b, kPe[i], e->pe_learn[i]);
  }
  build_mlp3(b, "prompt_encoder.motion_pred.state_encoder", {cfg->prompt_dim, D, D}, false, e->mlp_prompt);
  if (cfg->obs_fusion_mlp) build_mlp3(b, "scene_encoder.obs_update_mlp", {2 * D, D, D}, false, e->mlp_obs_fuse);
  if (cfg->goal_pred_k > 0) {
    build_mlp3(b, "decoder.goal_prob_head", {D, D / 2, cfg->goal_pred_k}, false, e->mlp_goal_prob);
    build_mlp3(b, "decoder.goal_point_head", {D, D / 2, 2 * cfg->goal_pred_k}, false, e->mlp_goal_point);
  }
  const std::string pa = "policy.act_decoder";
  if (!cfg->no_reconst_pred) build_mlp3(b, pa + ".pred_mlp", {D, D, D / 2, 2}, false, e->mlp_pred);
  // (TRAJ.PRED_MODE 'mlp': the head's last Linear carries all K modes, and there are no anchors / CG_decode blocks)
  const int head_out = cfg->target_steps * cfg->state_dim * (cfg->k_pred_mlp ? cfg->motion_k : 1);
  build_mlp3(b, pa + ".motion_head", {D, D, D / 2, head_out}, false, e->head.motion);
  if (!cfg->k_pred_mlp) b.plain(&e->head.anchors, pa + ".motion_anchors.weight", (int64_t)cfg->motion_k * cfg->num_agent_types * D);
  {
    // K-major copies of the motion head (reference MLP [128,128,64,out]: seq 0 Lin,1 LN,3 Lin,4 LN,6 Lin)
    const int OUT = head_out;
    const std::string mh = pa + ".motion_head.mlp.";
    b.fragments(&e->head.m0F, mh + "0.weight", D, D, 0, D);
    b.fragments(&e->head.m1F, mh + "3.weight", D / 2, D, 0, D);
    b.fragments(&e->head.m0Q, mh + "0.weight", D, D, 0, D, true);
    b.fragments(&e->head.m1Q, mh + "3.weight", D / 2, D, 0, D, true);
    b.transposed(&e->head.m0t, mh + "0.weight", D, D);
    b.plain(&e->head.m0b, mh + "0.bias", D);
    b.plain(&e->head.m0lnw, mh + "1.weight", D);
    b.plain(&e->head.m0lnb, mh + "1.bias", D);
    b.transposed(&e->head.m1t, mh + "3.weight", D / 2, D);
    b.plain(&e->head.m1b, mh + "3.bias", D / 2);
    b.plain(&e->head.m1lnw, mh + "4.weight", D / 2);
    b.plain(&e->head.m1lnb, mh + "4.bias", D / 2);
    const float* w2 = b.get(mh + "6.weight", (int64_t)OUT * (D / 2));
    const float* b2 = b.get(mh + "6.bias", OUT);
    if (w2 && b2) {
      // (zero-padded to 128 output columns = 8 MFMA n-tiles: 50 with the demo's 10 x 5, 80 with PRED_GMM's 10 x 8)
      std::vector<float> t((size_t)64 * 128, 0.f), bb(128, 0.f);
      for (int k = 0; k < 64; ++k)
        for (int n = 0; n < OUT && n < 128; ++n) t[(size_t)k * 128 + n] = w2[(size_t)n * 64 + k];
      for (int n = 0; n < OUT && n < 128; ++n) bb[n] = b2[n];
      b.slot(&e->head.m2t, b.put(t));
      std::vector<float> w2p((size_t)128 * 64, 0.f);   // torch layout [out 128 (zero rows past OUT)][in 64]
      for (int n = 0; n < OUT && n < 128; ++n) std::copy(w2 + (size_t)n * 64, w2 + (size_t)(n + 1) * 64, w2p.begin() + (size_t)n * 64);
      b.fragments_raw(&e->head.m2F, w2p.data(), 128, 64, 0, 64);
      b.fragments_raw(&e->head.m2Q, w2p.data(), 128, 64, 0, 64, true);
      b.slot(&e->head.m2b, b.put(bb));
    }
  }
  for (int i = 0; i < 3 && !cfg->k_pred_mlp; ++i) {
    const std::string q = pa + ".CG_decode.CGs." + std::to_string(i) + ".MLP.";
    b.fragments(&e->head.cgF[i], q + "0.weight", D, D, 0, D);
    b.fragments(&e->head.cgQ[i], q + "0.weight", D, D, 0, D, true);
    b.transposed(&e->head.cgWt[i], q + "0.weight", D, D);
    b.plain(&e->head.cgb[i], q + "0.bias", D);
    b.plain(&e->head.cglnw[i], q + "1.weight", D);
    b.plain(&e->head.cglnb[i], q + "1.bias", D);
  }
  const std::string ct = "condition_transformers.policy_decoder.condition_encoders";
  build_mlp3(b, ct + ".goal.goal_encoder", {2, D, D}, true, e->cond.goal);
  {
    std::vector<float> tags((size_t)11 * D, 0.f);
    for (int t = 0; t < 11; ++t) {
      const float* s = b.get(ct + ".v_action_tag.tag_encoder." + kTagNames[t], D);
      if (s) std::copy(s, s + D, tags.begin() + (size_t)t * D);
    }
    b.slot(&e->cond.tag_emb, b.put(tags));
    if (cfg->v2v_tag_mask) {   // V2V_MotionTagEncoder: [2 D] per used tag (source half | target half), V2V_MotionTag order
      static const char* kV2V[5] = {"Following", "ParallelDriving", "Merging", "ByPassing", "Overtaking"};
      std::vector<float> v2v((size_t)5 * 2 * D, 0.f);
      for (int t = 0; t < 5; ++t)
        if (cfg->v2v_tag_mask & (1 << t)) {
          const float* s_ = b.get(ct + ".v2v_tag.tag_encoder." + kV2V[t], 2 * D);
          if (s_) std::copy(s_, s_ + 2 * D, v2v.begin() + (size_t)t * 2 * D);
        }
      b.slot(&e->cond.v2v_emb, b.put(v2v));
    }
  }
  b.plain(&e->cond.div32, "const.fourier_div32", 32);
  b.plain(&e->cond.div64, "const.fourier_div64", 64);
  b.plain(&e->cond.div128, "const.fourier_div128", 128);
  if (!b.err.empty()) {
    ps_destroy(e);   // releases whatever was built so far
    return fail(PS_E_WEIGHT, b.err);
  }
  // upload the arena and rebase every recorded pointer slot
  const size_t nfl = e->arena_h.size();
  if (hipMalloc((void**)&e->arena_d, nfl * sizeof(float)) != hipSuccess) {
    ps_destroy(e);   // releases whatever was built so far
    return fail(PS_E_HIP, "hipMalloc(weights) failed");
  }
  if (hipMemcpy(e->arena_d, e->arena_h.data(), nfl * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    ps_destroy(e);   // releases whatever was built so far
    return fail(PS_E_HIP, "hipMemcpy(weights) failed");
  }
  for (const float** p : b.ptrs) {
    const size_t off = reinterpret_cast<size_t>(*p) - 1;
    *p = e->arena_d + off;
  }
  e->div32 = e->cond.div32;
  e->src.clear();
  // device copy of all attention layers for the kv-projection kernel
  std::vector<AttnW> all;
  auto add = [&](std::vector<AttnW>& v, int& base) {
    base = (int)all.size();
    all.insert(all.end(), v.begin(), v.end());
  };
  add(e->a2a, e->L_a2a);
  add(e->s2s, e->L_s2s);
  add(e->p2p, e->L_p2p);
  add(e->s2p, e->L_s2p);
  add(e->a2p, e->L_a2p);
  add(e->m2p, e->L_m2p);
  add(e->cnd, e->L_cnd);
  e->all_layers = all;
  if (hipMalloc((void**)&e->d_layers, std::max<size_t>(1, all.size()) * sizeof(AttnW)) != hipSuccess ||
      hipMemcpy(e->d_layers, all.data(), all.size() * sizeof(AttnW), hipMemcpyHostToDevice) != hipSuccess) {
    ps_destroy(e);   // releases whatever was built so far
    return fail(PS_E_HIP, "layer table upload failed");
  }
  if ((e->stream = engine_stream_acquire(e->cfg.device)) == nullptr || hipEventCreate(&e->ev0) != hipSuccess ||
      hipEventCreate(&e->ev1) != hipSuccess) {
    ps_destroy(e);   // releases whatever was built so far
    return fail(PS_E_HIP, "stream/event creation failed");
  }
  e->stage[0].device = e->stage[1].device = e->cfg.device;
  // the chain kernel may use up to ~140 KiB of dynamic LDS
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kv_proj), hipFuncAttributeMaxDynamicSharedMemorySize, (int)KV_LDS_BYTES);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pointnet_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PN_LDS_BYTES);
#define PS_RT_ATTR(K) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RT_LDS_BYTES)
  PS_RT_ATTR((k_pointnet_rt<1, 4>)); PS_RT_ATTR((k_pointnet_rt<1, 8>)); PS_RT_ATTR((k_pointnet_rt<1, 16>)); PS_RT_ATTR((k_pointnet_rt<2, 8>));
  PS_RT_ATTR((k_pointnet_rt<2, 16>)); PS_RT_ATTR((k_pointnet_rt<3, 4>)); PS_RT_ATTR((k_pointnet_rt<3, 8>)); PS_RT_ATTR((k_pointnet_rt<4, 8>));
  PS_RT_ATTR((k_pointnet_rt<5, 4>));
  PS_RT_ATTR(k_node_pre_rt<1>); PS_RT_ATTR(k_node_pre_rt<2>);
#ifdef PS_EXPERIMENTS   // (cross-check builds only, round 5: the 3-tile node halves spill 292 - 656 B per lane and k_edge16 is never selected)
  PS_RT_ATTR(k_node_pre_rt<3>); PS_RT_ATTR(k_node_post_rt<3>);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_edge16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c16_edge_lds_bytes());
#endif
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_edge_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ER_LDS_BYTES);
  PS_RT_ATTR(k_policy_head_rt);
  PS_RT_ATTR(k_node_post_rt<1>); PS_RT_ATTR(k_node_post_rt<2>);
#undef PS_RT_ATTR
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pe_learn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL_LDS_BYTES);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_node<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ND_LDS_BYTES);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_node<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ND_LDS_BYTES);
#define PS_C16_ATTR(NWW, POL, ONE) \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain16<NWW, POL, ONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c16_lds_bytes<NWW>())
  PS_C16_ATTR(8, true, true); PS_C16_ATTR(8, true, false); PS_C16_ATTR(8, false, true); PS_C16_ATTR(8, false, false);
#undef PS_C16_ATTR
#define PS_ATTR(TT, KRR, POL) \
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_chain<TT, 4, KRR, false, POL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
  PS_ATTR(1, 3, true); PS_ATTR(2, 3, true); PS_ATTR(4, 3, true);
  PS_ATTR(1, 3, false); PS_ATTR(2, 3, false); PS_ATTR(4, 3, false);
  PS_ATTR(1, 4, false); PS_ATTR(2, 4, false); PS_ATTR(4, 4, false);
#undef PS_ATTR
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_chain<1, 4, 3, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_chain<1, 4, 3, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  // (round 6: the weight uploads above went through the null stream, which is not ordered against the engine's non-blocking stream --
  // the engine's first launch must find them complete whatever the runtime's hipMemcpy waits for: drain the device once, here)
  if (hipDeviceSynchronize() != hipSuccess) { ps_destroy(e); return fail(PS_E_HIP, "device synchronisation after the weight upload failed"); }
  *out = e;
  return PS_OK;
}

extern "C" void ps_destroy(ps_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  (void)hipDeviceSynchronize();
  // (DevBuf members free themselves when the engine is deleted; the explicit releases keep the order: buffers before the stream)
  e->d_map_input.release(); e->d_obs_input.release(); e->d_prompt.release(); e->d_fut.release();
  e->d_is_policy.release(); e->d_tok_live.release(); e->d_live0.release(); e->d_obs_in_mask.release(); e->d_fut_mask.release();
  e->d_obs_mask_rows.release();
  e->d_fut_pos.release(); e->d_fut_head.release();
  e->d_map_mask.release(); e->d_obs_mask.release();
  e->d_map_rows.release(); e->d_agent_rows.release(); e->d_tok_scene.release(); e->d_agent_type.release();
  e->d_r_map.release(); e->d_r_agent.release(); e->d_r_zero.release();
  e->d_tok.release(); e->d_tok_pos.release(); e->d_tok_ori.release(); e->d_init_pos.release(); e->d_init_head.release();
  e->d_cur_pos.release(); e->d_cur_ori.release(); e->d_prompt_pos.release(); e->d_prompt_ori.release();
  e->d_xp.release(); e->d_emd.release(); e->d_xc.release(); e->d_fused.release(); e->d_obs_in.release();
  e->d_static_in.release(); e->d_kv.release(); e->d_kv_s2p.release(); e->d_kv_m2p.release(); e->d_kv_a2p.release();
  e->d_traj.release(); e->d_vel.release(); e->d_motion.release(); e->d_reconst.release(); e->d_choice.release(); e->d_world.release();
  e->d_kh.release(); e->d_kh_s2p.release(); e->d_kh_m2p.release(); e->d_kh_a2p.release();
  for (EdgeSet* s : {&e->e_a2a, &e->e_s2s, &e->e_p2p, &e->e_s2p, &e->e_a2p, &e->e_m2p, &e->e_cnd, &e->e_ua, &e->e_um}) {
    s->cnt.release(); s->eoff.release(); s->esrc.release(); s->edst.release(); s->toff.release(); s->tdst.release(); s->rtA.release(); s->rtT.release(); s->geo.release();
  }
  e->d_steps.release(); e->d_ent_off.release(); e->d_ent_type.release(); e->d_ent_val.release(); e->d_cond_edges.release();
  e->d_drag_in.release(); e->d_drag_emd.release(); e->d_drag_mask.release();
  e->d_obs_new.release(); e->d_kv_um.release(); e->d_kh_um.release();
  e->io_q.release(); e->io_qt.release(); e->io_cq.release(); e->io_ar.release(); e->io_av.release(); e->io_l.release();
  e->io_s.release(); e->io_g.release(); e->io_m.release();
  destroy_graph(e);
  e->stage[0].release();
  e->stage[1].release();
  if (e->arena_d) (void)hipFree(e->arena_d);
  if (e->d_layers) (void)hipFree(e->d_layers);
  for (auto ev : e->pev) if (ev) (void)hipEventDestroy(ev);
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->stream) engine_stream_release(e->cfg.device, e->stream);   // (drained by the hipDeviceSynchronize above; pooled, not destroyed)
  delete e;
}

// ------------------------------------------------------------------------------------------ scene upload
namespace {

template <class T>
int upload(DevBuf<T>& b, const T* h, size_t n, hipStream_t s) {
  if (b.ensure(n)) return -1;
  if (n == 0) return 0;
  if (g_stage) return g_stage->stage(b.p, h, n * sizeof(T), s);   // through the engine's arena: delivered when the setter ends (PinStage)
  return hipMemcpyAsync(b.p, h, n * sizeof(T), hipMemcpyHostToDevice, s) == hipSuccess ? 0 : -1;
}


// k_chain16 gathers geometry records, k rows and v rows through raw buffer descriptors with 32-bit byte offsets (edge * 32,
// source row * 512, source row * 1024 + 512; ps_chain16.h): beyond 2^31 bytes a buffer load returns 0 silently, so sizes that
// would get there are refused here.  token_rows: the largest source-row index a step can name (+ 1).
int check_c16_offsets(size_t token_rows, std::initializer_list<size_t> edge_caps) {
  constexpr size_t LIM = (size_t)1 << 31;
  if (token_rows * 1024 + 512 >= LIM)
    return fail(PS_E_ARG, "too many token rows for the fused chain's 32-bit gather offsets (rows * 1024 bytes must stay below 2^31: < 2 097 151 rows per engine)");
  for (size_t cap : edge_caps)
    if ((cap + 1) * 32 >= LIM)
      return fail(PS_E_ARG, "an edge set's capacity exceeds the fused chain's 32-bit record offsets (edges * 32 bytes must stay below 2^31: < 67 108 863 edges per set)");
  return 0;
}
// up to this many queries a search with geometry records is ONE launch of a workgroup per query (k_radius_geo, launch_radius below)
constexpr int SEARCH_WG_MAX_Q = 256;
int edge_alloc(EdgeSet& s, int nq, size_t cap_edges, int maxdeg, hipStream_t st) {
  s.nq = nq;
  s.cap_edges = cap_edges;
  s.maxdeg = std::max(1, maxdeg);
  if (s.cnt.ensure(nq + 1) || s.eoff.ensure(nq + 1) || s.toff.ensure(nq + 1) || s.tdst.ensure(cap_edges / 32 + (size_t)nq + 1) ||
      s.esrc.ensure(cap_edges + 1) ||
      s.edst.ensure(cap_edges + 1) ||
      s.rtA.ensure((cap_edges / 32 + (size_t)nq + 1) * 8192) || s.rtT.ensure((cap_edges / 32 + (size_t)nq + 1) * 8192) ||
      s.geo.ensure(cap_edges + 1))
    return -1;
  // The one-launch search's flags (k_radius_geo): zero before the set's first search, kept zero by the kernel itself.
  // Round 6: zeroed by a KERNEL ON THE ENGINE'S STREAM, for every new scene of a size that takes the one-launch search.  Until now a
  // (re)allocated block was cleared with hipMemset -- which runs on the null stream, is not ordered against the engine's NON-BLOCKING
  // stream and does not wait for the host either: with the GPU shared by eight processes it could land after the set's first searches had
  // started, wipe published counts and the done counter in mid-flight, and leave the next replay stale flags -- wrong CSR offsets, edge
  // lists out of their ranges, a GPU memory fault (rank 0 of the 8-processes-on-one-GPU bench test, 1 run in 4: profiles/r06_k_*).
  if (s.sync.ensure(2 * (size_t)nq + 8)) return -1;   // 64 bits per query + the counter
  if (nq <= SEARCH_WG_MAX_Q && dev_zero(st, s.sync.p, (2 * (size_t)nq + 8) * sizeof(int))) return -1;
  return 0;
}

}  // namespace

extern "C" int ps_set_scene(ps_engine* e, int32_t B, int32_t M, int32_t P, int32_t N, const float* map_input,
                            const uint8_t* map_mask, const float* map_pos, const float* map_head, const float* obs_input,
                            const uint8_t* obs_mask, const float* obs_pos, const float* obs_head, const float* prompt,
                            const uint8_t* prompt_mask, const int32_t* agent_type, const float* prompt_pos,
                            const float* prompt_head) {
  if (!e) return fail(PS_E_ARG, "null engine");
  if (B < 1 || M < 1 || N < 1 || P < 1 || P > 32) return fail(PS_E_ARG, "need B,M,N >= 1 and 1 <= P <= 32");
  const ps_config& c = e->cfg;
  HIPCHK(hipSetDevice(c.device));
  const int Hs = c.hist_steps, Od = c.obs_dim;
  // whatever happens below, the previous scene is gone: a caller that catches an error must not be able to replay a
  // graph (or read results) over buffers this call has already resized
  e->have_scene = e->encoded = e->generated = e->reset = false;
  drop_graph(e);
  // uploads go through the engine's pinned arena and this call ends without a stream synchronisation (PinStage)
  e->stage_cur ^= 1;   // (the other arena may still feed the uploads of a batch whose rollout is in flight)
  if (e->stage[e->stage_cur].begin(e->stream)) return fail(PS_E_HIP, "upload staging: the previous uploads did not complete");
  StageScope stage_scope(&e->stage[e->stage_cur], e->stream);
  struct DeclaredGuard {   // ps_declare_agent_rows is consumed by this call on EVERY exit path
    std::vector<uint8_t> rows;
    std::vector<uint8_t>& ref;
    explicit DeclaredGuard(std::vector<uint8_t>& r) : rows(r), ref(r) { ref.clear(); }
  } declared_guard(e->declared_rows);
  const std::vector<uint8_t>& declared_rows = declared_guard.rows;
  e->replicas = e->replicas_next;
  if (e->replicas > 1 && B != 1) return fail(PS_E_ARG, "ps_set_replicas: the replicated batch holds ONE scene");
  const int Bi = e->replicas > 1 ? e->replicas : B;   // scenes the kernels see
  e->B = B; e->M = M; e->P = P; e->N = N;
  e->map_rows.clear(); e->agent_rows.clear(); e->agent_scene.clear(); e->map_scene.clear();
  e->is_policy_h.clear();
  e->live0_h.clear();
  const bool declared = declared_rows.size() == (size_t)B * N;
  if (!declared_rows.empty() && !declared)
    return fail(PS_E_ARG, "ps_declare_agent_rows was called with another [B, N] than this ps_set_scene");
  e->moff.assign(B + 1, 0); e->aoff.assign(B + 1, 0);
  e->maxA_scene = e->maxM_scene = 0;
  for (int b = 0; b < B; ++b) {
    for (int m = 0; m < M; ++m) {
      bool any = false;
      for (int p = 0; p < P; ++p) any |= map_mask[((size_t)b * M + m) * P + p] != 0;
      if (any) { e->map_rows.push_back(b * M + m); e->map_scene.push_back(b); }
    }
    e->moff[b + 1] = (int)e->map_rows.size();
    for (int n = 0; n < N; ++n) {
      bool any = false;
      for (int s = 0; s < Hs && !any; ++s) {
        bool all = true;
        const uint8_t* mk = obs_mask + (((size_t)b * N + n) * Hs + s) * Od;
        for (int f = 0; f < Od; ++f) all &= mk[f] != 0;
        any |= all;
      }
      const bool pol = prompt_mask[(size_t)b * N + n] != 0;
      if (pol && !any)
        return fail(PS_E_ARG, "a policy agent must be observed: its prompt sits in the slot of its history (scene " +
                                  std::to_string(b) + ", slot " + std::to_string(n) + ")");
      if (any || (declared && declared_rows[(size_t)b * N + n])) {   // a declared row without history enters later
        e->live0_h.push_back(any ? 1 : 0);
        e->is_policy_h.push_back(pol ? 1 : 0);
        const int ty = agent_type[(size_t)b * N + n];
        if (ty < 1 || ty > c.num_agent_types) return fail(PS_E_ARG, "agent_type outside 1..num_agent_types");
        e->agent_rows.push_back(b * N + n);
        e->agent_scene.push_back(b);
      }
    }
    e->aoff[b + 1] = (int)e->agent_rows.size();
    e->maxA_scene = std::max(e->maxA_scene, e->aoff[b + 1] - e->aoff[b]);
    e->maxM_scene = std::max(e->maxM_scene, e->moff[b + 1] - e->moff[b]);
  }
  // begin / end of every scene's map tokens and agent rows
  std::vector<int> mbeg(Bi), mend(Bi);
  if (e->replicas > 1) {   // replica r: the same input slots again, its own agent rows, the shared map tokens
    const int A1 = (int)e->agent_rows.size();
    for (int r = 1; r < e->replicas; ++r)
      for (int i = 0; i < A1; ++i) {
        e->agent_rows.push_back(e->agent_rows[i]);
        e->agent_scene.push_back(r);
        e->live0_h.push_back(e->live0_h[i]);
        e->is_policy_h.push_back(e->is_policy_h[i]);
      }
    e->aoff.assign(Bi + 1, 0);
    for (int r = 0; r <= Bi; ++r) e->aoff[r] = r * A1;
    for (int r = 0; r < Bi; ++r) { mbeg[r] = 0; mend[r] = (int)e->map_rows.size(); }
    e->Ap = A1;
  } else {
    for (int b = 0; b < B; ++b) { mbeg[b] = e->moff[b]; mend[b] = e->moff[b + 1]; }
    e->Ap = (int)e->agent_rows.size();
  }
  e->agent_slot.resize(e->agent_rows.size());
  for (size_t i = 0; i < e->agent_rows.size(); ++i)
    e->agent_slot[i] = e->replicas > 1 ? e->agent_scene[i] * N + e->agent_rows[i] : e->agent_rows[i];
  if (e->maxA_scene + e->maxM_scene > 64 * KNN_SLOTS)
    return fail(PS_E_ARG, "more than 2560 tokens in one scene (knn candidate registers)");
  const int Mv = e->Mv = (int)e->map_rows.size();
  const int A = e->A = (int)e->agent_rows.size();
  const int Ap = e->Ap;
  if (A == 0) return fail(PS_E_ARG, "no valid agents");
  e->have_dead0 = false;
  for (int v : e->live0_h) e->have_dead0 |= v == 0;
  e->n_policy = 0;
  for (int v : e->is_policy_h) e->n_policy += v;
  if (e->n_policy == 0) return fail(PS_E_ARG, "no policy agent (prompt_mask is empty)");
  e->all_policy = e->n_policy == A;
  e->have_log = false;
  hipStream_t st = e->stream;
  // raw inputs
  if (upload(e->d_map_input, map_input, (size_t)B * M * P * c.map_dim, st) || upload(e->d_map_mask, map_mask, (size_t)B * M * P, st) ||
      upload(e->d_obs_input, obs_input, (size_t)B * N * Hs * Od, st) || upload(e->d_obs_mask, obs_mask, (size_t)B * N * Hs * Od, st) ||
      upload(e->d_prompt, prompt, (size_t)B * N * c.prompt_dim, st) || upload(e->d_map_rows, e->map_rows.data(), (size_t)Mv, st) ||
      upload(e->d_agent_rows, e->agent_rows.data(), (size_t)A, st))
    return fail(PS_E_HIP, "input upload failed");
  // compact token geometry: [map tokens ; agent tokens], scene-major inside each part
  std::vector<float> pos((size_t)(Mv + A) * 2), ori(Mv + A), ipos((size_t)A * 2), ihead(A), ppos((size_t)A * 2), pori(A);
  std::vector<int> scene(Mv + A), atype(A);
  for (int i = 0; i < Mv; ++i) {
    pos[2 * i] = map_pos[2 * (size_t)e->map_rows[i]];
    pos[2 * i + 1] = map_pos[2 * (size_t)e->map_rows[i] + 1];
    ori[i] = map_head[e->map_rows[i]];
    scene[i] = e->map_scene[i];
  }
  for (int i = 0; i < A; ++i) {
    const size_t r = e->agent_rows[i];
    pos[2 * (Mv + i)] = ipos[2 * i] = obs_pos[2 * r];
    pos[2 * (Mv + i) + 1] = ipos[2 * i + 1] = obs_pos[2 * r + 1];
    ori[Mv + i] = ihead[i] = obs_head[r];
    ppos[2 * i] = prompt_pos ? prompt_pos[2 * r] : obs_pos[2 * r];
    ppos[2 * i + 1] = prompt_pos ? prompt_pos[2 * r + 1] : obs_pos[2 * r + 1];
    pori[i] = prompt_head ? prompt_head[r] : obs_head[r];
    scene[Mv + i] = e->agent_scene[i];
    atype[i] = agent_type[r];
  }
  std::vector<int> r_map(2 * Bi), r_agent(2 * Bi), r_zero(2 * Bi, 0);
  for (int b = 0; b < Bi; ++b) {
    r_map[2 * b] = mbeg[b]; r_map[2 * b + 1] = mend[b];
    r_agent[2 * b] = Mv + e->aoff[b]; r_agent[2 * b + 1] = Mv + e->aoff[b + 1];
  }
  if (upload(e->d_tok_pos, pos.data(), pos.size(), st) || upload(e->d_tok_ori, ori.data(), ori.size(), st) ||
      upload(e->d_init_pos, ipos.data(), ipos.size(), st) || upload(e->d_init_head, ihead.data(), ihead.size(), st) ||
      upload(e->d_prompt_pos, ppos.data(), ppos.size(), st) || upload(e->d_prompt_ori, pori.data(), pori.size(), st) ||
      e->d_cur_pos.ensure(ppos.size()) || e->d_cur_ori.ensure(pori.size()) ||
      upload(e->d_tok_scene, scene.data(), scene.size(), st) || upload(e->d_agent_type, atype.data(), atype.size(), st) ||
      upload(e->d_r_map, r_map.data(), r_map.size(), st) || upload(e->d_r_agent, r_agent.data(), r_agent.size(), st) ||
      upload(e->d_r_zero, r_zero.data(), r_zero.size(), st))
    return fail(PS_E_HIP, "geometry upload failed");
  // (the uploads copy out of these vectors into the pinned arena before they return)
  std::vector<float> stat;
  std::vector<uint8_t> mrows;
  std::vector<int> off, tof, tds, off2, tof2, tds2;
  // static observation columns (extent, type, time one-hot) default to init_obs
  {
    stat.assign((size_t)A * Hs * Od, 0.f);
    for (int i = 0; i < A; ++i)
      std::memcpy(&stat[(size_t)i * Hs * Od], obs_input + (size_t)e->agent_rows[i] * Hs * Od, sizeof(float) * Hs * Od);
    if (upload(e->d_static_in, stat.data(), stat.size(), st)) return fail(PS_E_HIP, "static obs upload failed");
  }
  e->have_fut = false;
  {
    const size_t arow = (size_t)c.hist_steps * c.obs_dim;
    mrows.assign((size_t)A * arow, 0);
    for (int i = 0; i < A; ++i) std::memcpy(&mrows[(size_t)i * arow], obs_mask + (size_t)e->agent_rows[i] * arow, arow);
    if (upload(e->d_is_policy, e->is_policy_h.data(), (size_t)A, st) || e->d_tok_live.ensure((size_t)A) ||
        upload(e->d_live0, e->live0_h.data(), (size_t)A, st) ||
        e->d_obs_in_mask.ensure((size_t)A * arow) || upload(e->d_obs_mask_rows, mrows.data(), mrows.size(), st))
      return fail(PS_E_HIP, "policy-flag upload failed");
  }
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq;
  e->stride_steps = Hs + R * c.replan_freq;
  const int L6 = std::max({c.scene_layers, c.dec_layers, c.pol_layers, c.cond_layers, 1});
  if (e->d_tok.ensure((size_t)(Mv + A) * D) || e->d_xp.ensure((size_t)A * D) || e->d_emd.ensure((size_t)A * D) ||
      e->d_xc.ensure((size_t)A * D) || e->d_fused.ensure((size_t)A * D) || e->d_obs_in.ensure((size_t)A * Hs * Od) ||
      e->d_kv.ensure((size_t)(Mv + A) * 256) || e->d_kv_s2p.ensure((size_t)L6 * (Mv + A) * 256) ||
      e->d_kv_m2p.ensure((size_t)L6 * std::max(Mv, 1) * 256) || e->d_kv_a2p.ensure((size_t)L6 * A * 256) ||
      e->d_kh.ensure((size_t)(Mv + A) * 256) || e->d_kh_s2p.ensure((size_t)L6 * (Mv + A) * 256) ||
      e->d_kh_m2p.ensure((size_t)L6 * std::max(Mv, 1) * 256) || e->d_kh_a2p.ensure((size_t)L6 * A * 256) ||
      e->d_traj.ensure((size_t)A * e->stride_steps * 4) || e->d_vel.ensure((size_t)A * e->stride_steps * 2) ||
      e->d_motion.ensure((size_t)R * A * c.motion_k * c.target_steps * c.state_dim) || e->d_reconst.ensure((size_t)A * 2) ||
      e->d_choice.ensure((size_t)R * A) || e->d_goal_prob.ensure((size_t)A * std::max(1, c.goal_pred_k)) ||
      e->d_goal_point.ensure((size_t)A * 2 * std::max(1, c.goal_pred_k)))
    return fail(PS_E_HIP, "device allocation failed");
  if (dev_zero(st, e->d_choice.p, sizeof(int) * (size_t)R * A)) return fail(PS_E_HIP, "device fill launch failed");   // mode 0 until ps_set_mode_choice
  e->have_noise = false;                                                        // no action noise until ps_set_action_noise
  // ---- edge sets: capacities from worst-case degrees
  auto mn = [](int a, int b) { return a < b ? a : b; };
  const int tokS = e->maxA_scene + e->maxM_scene;
  int d_a2a = mn(c.agent_knn, e->maxA_scene), d_s2s = mn(c.scene_knn, tokS);
  // radius_graph(loop=False) scans cap+1 matches and then drops the self match: a query whose own
  // index is not among the first cap+1 keeps all cap+1 of them
  int d_p2p = mn(c.dec_max_neigh + 1, std::max(1, e->maxA_scene - 1)), d_s2p = mn(c.dec_max_neigh, tokS);
  int d_a2p = mn(c.pol_max_neigh, e->maxA_scene), d_m2p = mn(c.pol_max_neigh, std::max(1, e->maxM_scene));
  if (c.obs_fusion_mlp && e->d_obs_new.ensure((size_t)A * D)) return fail(PS_E_HIP, "device allocation failed");
  if (c.obs_attn_update) {
    const int d_ua = mn(c.scene_knn + 1, std::max(1, e->maxA_scene - 1)), d_um = mn(c.scene_knn, std::max(1, e->maxM_scene));
    if (edge_alloc(e->e_ua, A, (size_t)A * d_ua, d_ua, e->stream) || edge_alloc(e->e_um, A, (size_t)A * d_um, d_um, e->stream) ||
        e->d_kv_um.ensure((size_t)c.scene_layers * std::max(Mv, 1) * 256) || e->d_kh_um.ensure((size_t)c.scene_layers * std::max(Mv, 1) * 256))
      return fail(PS_E_HIP, "edge allocation failed");
  }
  if (check_c16_offsets((size_t)Mv + A, {(size_t)Ap * d_a2a, (size_t)(Mv + Ap) * d_s2s, (size_t)Ap * d_p2p, (size_t)Ap * d_s2p,
                                          (size_t)A * d_a2p, (size_t)A * d_m2p}))
    return PS_E_ARG;
  // (the scene encoder's and the generator's sets have Ap destination rows: with replicas they run once)
  if (edge_alloc(e->e_a2a, Ap, (size_t)Ap * d_a2a, d_a2a, e->stream) || edge_alloc(e->e_s2s, Mv + Ap, (size_t)(Mv + Ap) * d_s2s, d_s2s, e->stream) ||
      edge_alloc(e->e_p2p, Ap, (size_t)Ap * d_p2p, d_p2p, e->stream) || edge_alloc(e->e_s2p, Ap, (size_t)Ap * d_s2p, d_s2p, e->stream) ||
      edge_alloc(e->e_a2p, A, (size_t)A * d_a2p, d_a2p, e->stream) || edge_alloc(e->e_m2p, A, (size_t)A * d_m2p, d_m2p, e->stream) ||
      edge_alloc(e->e_cnd, A, (size_t)A, 1, e->stream))
    return fail(PS_E_HIP, "edge allocation failed");
  {   // split-path exchange buffers for the largest destination set (allocated here, never inside a captured rollout)
    EdgeIO io_;
    if (io_for(e, std::max(Mv + A, 8 * A), io_)) return fail(PS_E_HIP, "split-path buffers");
  }
  // closed-form CSR offsets of the knn graphs (every query gets min(k, scene size) neighbours)
  {
    std::vector<int> nlive(Bi, 0);   // agents in the scene at the initial step (the kNN candidates)
    for (int i = 0; i < A; ++i) nlive[e->agent_scene[i]] += e->live0_h[i];
    off.assign(Ap + 1, 0);
    tof.assign(Ap + 1, 0);
    for (int i = 0; i < Ap; ++i) {
      const int b = e->agent_scene[i];
      const int dg = mn(c.agent_knn, nlive[b]);
      off[i + 1] = off[i] + dg;
      tof[i + 1] = tof[i] + (dg + 31) / 32;
      tds.insert(tds.end(), (dg + 31) / 32, i);
    }
    if (upload(e->e_a2a.eoff, off.data(), off.size(), st) || upload(e->e_a2a.toff, tof.data(), tof.size(), st) ||
        upload(e->e_a2a.tdst, tds.data(), tds.size(), st))
      return fail(PS_E_HIP, "upload failed");
    e->edge_counts[0] = (float)off[Ap];
    off2.assign(Mv + Ap + 1, 0);
    tof2.assign(Mv + Ap + 1, 0);
    for (int i = 0; i < Mv + Ap; ++i) {
      const int b = scene[i];
      const int ns = nlive[b] + (mend[b] - mbeg[b]);
      const int dg = mn(c.scene_knn, ns);
      off2[i + 1] = off2[i] + dg;
      tof2[i + 1] = tof2[i] + (dg + 31) / 32;
      tds2.insert(tds2.end(), (dg + 31) / 32, i);
    }
    if (upload(e->e_s2s.eoff, off2.data(), off2.size(), st) || upload(e->e_s2s.toff, tof2.data(), tof2.size(), st) ||
        upload(e->e_s2s.tdst, tds2.data(), tds2.size(), st))
      return fail(PS_E_HIP, "upload failed");
    e->edge_counts[1] = (float)off2[Mv + Ap];
  }
  // ---- chain step tables (device pointers are stable until the next ps_set_scene)
  e->h_steps.clear();
  auto push = [&](const AttnW& w, const float* kv, const _Float16* khl, EdgeSet& es) {
    ChainStep s;
    s.w = w;
    s.kv = kv;
    s.khl = khl;
    s.rtA = es.rtA.p;
    s.eoff = es.eoff.p;
    s.esrc = es.esrc.p;
    s.toff = es.toff.p;
    s.rtT = es.rtT.p;
    s.kr = (&es == &e->e_cnd || pe_of(e, &es)) ? 4 : 3;   // learnable rel-PE rows have no repeated columns to fold
    s.geo = es.geo.p;
    e->h_steps.push_back(s);
  };
  e->step_a2a = (int)e->h_steps.size();
  for (int i = 0; i < c.scene_layers; ++i) push(e->a2a[i], e->d_kv.p, e->d_kh.p, e->e_a2a);
  e->step_s2s = (int)e->h_steps.size();
  for (int i = 0; i < c.scene_layers; ++i) push(e->s2s[i], e->d_kv.p, e->d_kh.p, e->e_s2s);
  e->step_dec = (int)e->h_steps.size();
  for (int i = 0; i < c.dec_layers; ++i) {
    push(e->p2p[i], e->d_kv.p, e->d_kh.p, e->e_p2p);
    push(e->s2p[i], e->d_kv_s2p.p + (size_t)i * (Mv + Ap) * 256, e->d_kh_s2p.p + (size_t)i * (Mv + Ap) * 256, e->e_s2p);
  }
  e->step_cnd = (int)e->h_steps.size();
  for (int i = 0; i < c.cond_layers; ++i) push(e->cnd[i], e->d_kv.p, e->d_kh.p, e->e_cnd);
  e->step_pol = (int)e->h_steps.size();
  static const int abl = exp_env("PS_C16_ABL") ? atoi(exp_env("PS_C16_ABL")) : 0;   // experiments only (timing, wrong results): 1 = every policy layer reads layer 0's k | v
  for (int i0 = 0; i0 < c.pol_layers; ++i0) {
    const int i = (abl & 1) ? 0 : i0;
    // a2p edges carry GLOBAL agent rows (Mv + j); the kv buffer is agent-local -> bias the base by -Mv rows
    push(e->a2p[i0], e->d_kv_a2p.p + (size_t)i * A * 256 - (size_t)Mv * 256, e->d_kh_a2p.p + (size_t)i * A * 256 - (size_t)Mv * 256, e->e_a2p);
    push(e->m2p[i0], e->d_kv_m2p.p + (size_t)i * Mv * 256, e->d_kh_m2p.p + (size_t)i * Mv * 256, e->e_m2p);
  }
  e->step_upd = (int)e->h_steps.size();
  if (c.obs_attn_update)
    for (int i = 0; i < c.scene_layers; ++i) {   // _update_scene_emb_attn: a2a over the agents, then the s2s layer with (map -> agents) edges
      push(e->a2a[i], e->d_kv.p, e->d_kh.p, e->e_ua);
      push(e->s2s[i], e->d_kv_um.p + (size_t)i * Mv * 256, e->d_kh_um.p + (size_t)i * Mv * 256, e->e_um);
    }
  if (upload(e->d_steps, e->h_steps.data(), e->h_steps.size(), st)) return fail(PS_E_HIP, "step table upload failed");
  // no conditions until ps_set_conditions
  e->have_cond = false;
  e->n_cond_edges = 0;
  e->ents_gt.clear();
  e->ents_drag.clear();
  e->ents_pair.clear();
  e->n_drag = 0;
  e->cond_present_gt = e->cond_present_drag = e->cond_present_pair = false;
  if (e->stage[e->stage_cur].mark(st)) return fail(PS_E_HIP, "upload staging: event record failed");
  stage_scope.commit();
  e->have_scene = true;
  e->encoded = e->generated = e->reset = false;
  drop_graph(e);
  // keep host copies the later stages need
  return PS_OK;
}

extern "C" int ps_declare_agent_rows(ps_engine* e, int32_t B, int32_t N, const uint8_t* rows) {
  if (!e) return fail(PS_E_ARG, "null engine");
  e->declared_rows.clear();
  if (rows && B > 0 && N > 0) e->declared_rows.assign(rows, rows + (size_t)B * N);
  return PS_OK;
}

extern "C" int ps_set_prompt(ps_engine* e, const float* prompt, const float* prompt_pos, const float* prompt_head,
                             const int32_t* agent_type) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_set_prompt before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int A = e->A;
  std::vector<float> ppos((size_t)A * 2), pori(A);
  std::vector<int> atype(A);
  for (int i = 0; i < A; ++i) {
    const size_t r = e->agent_rows[i];
    ppos[2 * i] = prompt_pos[2 * r];
    ppos[2 * i + 1] = prompt_pos[2 * r + 1];
    pori[i] = prompt_head[r];
    atype[i] = agent_type[r];
    if (atype[i] < 1 || atype[i] > c.num_agent_types) return fail(PS_E_ARG, "agent_type outside 1..num_agent_types");
  }
  if (upload(e->d_prompt, prompt, (size_t)e->B * e->N * c.prompt_dim, e->stream) ||
      upload(e->d_prompt_pos, ppos.data(), ppos.size(), e->stream) || upload(e->d_prompt_ori, pori.data(), pori.size(), e->stream) ||
      upload(e->d_agent_type, atype.data(), atype.size(), e->stream))
    return fail(PS_E_HIP, "prompt upload failed");
  HIPCHK(hipStreamSynchronize(e->stream));
  e->generated = false;
  drop_graph(e);
  return PS_OK;
}

// Device form of the condition layers' graph (_construct_cond_edge_matrix + _pool_edges, condition_attns.py:114-188):
// the distinct (source -> destination) pairs that carry a condition entry -- self loops for unary conditions, s -> t and
// t -> s for binary ones -- as a CSR by destination with 32-edge tiles, and per edge the list of its entries (type, id,
// 3 floats) in the order goal, tags, drag points, binary tags.
static int rebuild_conditions(ps_engine* e) {
  const int A = e->Ap;   // (with replicas the condition layers run once, over the first replica's rows)
  struct Key { int dst, src; };
  std::vector<std::pair<Key, const ps_engine::CondEnt*>> all;
  for (const auto* list : {&e->ents_gt, &e->ents_drag, &e->ents_pair})
    for (const auto& en : *list) all.push_back({Key{en.agent, en.src < 0 ? en.agent : en.src}, &en});
  std::stable_sort(all.begin(), all.end(), [](const auto& x, const auto& y) {
    return x.first.dst != y.first.dst ? x.first.dst < y.first.dst : x.first.src < y.first.src;
  });
  std::vector<int> eoff(A + 1, 0), toff(A + 1, 0), esrc, tdst, ent_off(1, 0), ent_type;
  std::vector<float> ent_val;
  std::vector<CondEdge> edges;
  int maxdeg = 1;
  size_t i = 0;
  for (int a = 0; a < A; ++a) {
    int deg = 0;
    while (i < all.size() && all[i].first.dst == a) {
      const int src = all[i].first.src;
      size_t j_end = i;
      while (j_end < all.size() && all[j_end].first.dst == a && all[j_end].first.src == src) ++j_end;
      for (; i < j_end; ++i) {
        const auto* en = all[i].second;
        // the reference writes one plane per condition key by assignment (condition_attns.py:155-166): of several entries of one
        // key (type, id) on one edge the LAST one survives (the stable sort kept the entries' order) and counts once
        bool later = false;
        for (size_t j = i + 1; j < j_end && !later; ++j) later = all[j].second->type == en->type && all[j].second->id == en->id;
        // a binary key's plane is written in TWO passes (:155-162: every source half on (s, t), then every target half on (t, s)): the
        // target half of a REVERSED entry of the same tag overwrites a source half on this edge, wherever it stands in the entry order
        if (!later && en->type == 3 && (en->id & 1) == 0) {
          size_t j0 = i;
          while (j0 > 0 && all[j0 - 1].first.dst == a && all[j0 - 1].first.src == src) --j0;
          for (size_t j = j0; j < j_end && !later; ++j) later = all[j].second->type == 3 && all[j].second->id == en->id + 1;
        }
        if (later) continue;
        ent_type.push_back(en->type);
        ent_type.push_back(en->id);
        ent_val.insert(ent_val.end(), en->v, en->v + 3);
      }
      ent_off.push_back((int)ent_type.size() / 2);
      esrc.push_back(src);
      edges.push_back(CondEdge{src, a, (toff[a] + deg / 32) * 32 + (deg & 31)});
      ++deg;
    }
    eoff[a + 1] = (int)esrc.size();
    toff[a + 1] = toff[a] + (deg + 31) / 32;
    for (int t = toff[a]; t < toff[a + 1]; ++t) tdst.push_back(a);
    maxdeg = std::max(maxdeg, deg);
  }
  if (maxdeg > 2047) return fail(PS_E_ARG, "more than 2047 condition edges into one prompt");
  e->n_cond_edges = (int)esrc.size();
  e->n_cond_tiles = toff[A];
  e->have_cond = e->cond_present_gt || e->cond_present_drag || e->cond_present_pair;
  e->edge_counts[6] = (float)e->n_cond_edges;
  hipStream_t st = e->stream;
  StageScope stage_scope(&e->stage[e->stage_cur], e->stream);   // (appends to the arena ps_set_scene started; no stream synchronisation at the end)
  // (the set was sized for one self loop per prompt; binary conditions add edges: grow, and re-point the layers' steps)
  const _Float16 *oa = e->e_cnd.rtA.p, *ot = e->e_cnd.rtT.p;
  const int *oo = e->e_cnd.eoff.p, *os_ = e->e_cnd.esrc.p, *of = e->e_cnd.toff.p;
  e->e_cnd.maxdeg = maxdeg;
  if (e->e_cnd.rtA.ensure((size_t)(e->n_cond_tiles + 1) * 8192) || e->e_cnd.rtT.ensure((size_t)(e->n_cond_tiles + 1) * 8192) ||
      upload(e->e_cnd.eoff, eoff.data(), eoff.size(), st) || upload(e->e_cnd.toff, toff.data(), toff.size(), st) ||
      upload(e->e_cnd.esrc, esrc.data(), esrc.size(), st) || upload(e->e_cnd.tdst, tdst.data(), tdst.size(), st) ||
      upload(e->d_cond_edges, edges.data(), edges.size(), st) ||
      upload(e->d_ent_off, ent_off.data(), ent_off.size(), st) || upload(e->d_ent_type, ent_type.data(), ent_type.size(), st) ||
      upload(e->d_ent_val, ent_val.data(), ent_val.size(), st))
    return fail(PS_E_HIP, "condition upload failed");
  if (oa != e->e_cnd.rtA.p || ot != e->e_cnd.rtT.p || oo != e->e_cnd.eoff.p || os_ != e->e_cnd.esrc.p || of != e->e_cnd.toff.p) {
    for (int l = 0; l < e->cfg.cond_layers; ++l) {
      ChainStep& sp_ = e->h_steps[e->step_cnd + l];
      sp_.rtA = e->e_cnd.rtA.p; sp_.rtT = e->e_cnd.rtT.p; sp_.eoff = e->e_cnd.eoff.p; sp_.esrc = e->e_cnd.esrc.p; sp_.toff = e->e_cnd.toff.p;
    }
    if (upload(e->d_steps, e->h_steps.data(), e->h_steps.size(), st)) return fail(PS_E_HIP, "step table upload failed");
  }
  if (e->stage[e->stage_cur].mark(st)) return fail(PS_E_HIP, "upload staging: event record failed");
  stage_scope.commit();
  e->generated = false;
  drop_graph(e);
  return PS_OK;
}

extern "C" int ps_set_conditions(ps_engine* e, int32_t C_goal, const float* goal_input, const uint8_t* goal_mask,
                                 const int32_t* goal_pidx, int32_t C_tag, const float* tag_input, const uint8_t* tag_mask,
                                 const int32_t* tag_pidx) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_set_conditions before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  const int A = e->A, N = e->N;
  std::vector<ps_engine::CondEnt> ents;
  // GoalConditionEncoder emits its entry whenever the type has rows (condition_encoders.py:21-51); MotionTagEncoder
  // emits one per used tag that occurs in the input, whatever the mask says (:106-111)
  bool present = C_goal > 0 && goal_input;
  for (size_t i = 0; i < (size_t)e->B * (C_tag > 0 && tag_input ? C_tag : 0); ++i) {
    const int tag = (int)tag_input[3 * i];
    present |= tag >= 0 && tag <= 10;
  }
  e->cond_present_gt = present;
  // slot -> compact agent index
  std::vector<int> slot2a((size_t)e->B * N, -1);
  for (int i = 0; i < e->Ap; ++i) slot2a[e->agent_rows[i]] = i;
  for (int b = 0; b < e->B; ++b) {
    for (int c = 0; c < C_goal && goal_input; ++c) {
      const size_t i = (size_t)b * C_goal + c;
      if (!goal_mask[i]) continue;
      const int n = goal_pidx[i];
      if (n < 0 || n >= N || slot2a[(size_t)b * N + n] < 0) return fail(PS_E_ARG, "goal condition on an invalid prompt slot");
      ents.push_back({slot2a[(size_t)b * N + n], 0, 0, {goal_input[3 * i], goal_input[3 * i + 1], goal_input[3 * i + 2]}});
    }
    for (int c = 0; c < C_tag && tag_input; ++c) {
      const size_t i = (size_t)b * C_tag + c;
      if (!tag_mask[i]) continue;
      const int tag = (int)tag_input[3 * i];
      if (tag < 0 || tag > 10) continue;  // not a V_Action tag value: no entry (condition_encoders.py:94)
      const int n = tag_pidx[i];
      if (n < 0 || n >= N || slot2a[(size_t)b * N + n] < 0) return fail(PS_E_ARG, "tag condition on an invalid prompt slot");
      ents.push_back({slot2a[(size_t)b * N + n], 1, tag, {tag_input[3 * i], tag_input[3 * i + 1], tag_input[3 * i + 2]}});
    }
  }
  e->ents_gt.swap(ents);
  return rebuild_conditions(e);
}

extern "C" int ps_set_drag_points(ps_engine* e, int32_t C_drag, int32_t T, const float* drag_input, const uint8_t* drag_mask,
                                  const int32_t* drag_pidx) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_set_drag_points before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  if (C_drag <= 0 || !drag_input) {   // clear
    e->ents_drag.clear();
    e->n_drag = 0;
    e->cond_present_drag = false;
    return rebuild_conditions(e);
  }
  if (e->cfg.drag_mlp_layers <= 0) return fail(PS_E_ARG, "this engine was created without the drag-point encoder (drag_mlp_layers = 0)");
  if (T < 1 || T > 32) return fail(PS_E_ARG, "drag-point conditions carry 1..32 points");
  if (!drag_mask || !drag_pidx) return fail(PS_E_ARG, "drag_mask / drag_pidx missing");
  const int A = e->A, N = e->N;
  std::vector<int> slot2a((size_t)e->B * N, -1);
  for (int i = 0; i < e->Ap; ++i) slot2a[e->agent_rows[i]] = i;
  std::vector<ps_engine::CondEnt> ents;
  std::vector<float> pts;
  std::vector<uint8_t> pm;
  for (int b = 0; b < e->B; ++b)
    for (int c = 0; c < C_drag; ++c) {
      const size_t i = (size_t)b * C_drag + c;
      if (!drag_mask[i]) continue;
      const int n = drag_pidx[i];
      if (n < 0 || n >= N || slot2a[(size_t)b * N + n] < 0) return fail(PS_E_ARG, "drag-point condition on an invalid prompt slot");
      const int row = (int)ents.size();
      ents.push_back({slot2a[(size_t)b * N + n], 2, row, {0.f, 0.f, 0.f}});
      for (int t = 0; t < T; ++t) {   // a point is valid iff neither coordinate is NaN (condition_encoders.py:180)
        const float x = drag_input[(i * T + t) * 2], y = drag_input[(i * T + t) * 2 + 1];
        const bool ok = !(std::isnan(x) || std::isnan(y));
        pm.push_back(ok ? 1 : 0);
        pts.push_back(ok ? x : 0.f);
        pts.push_back(ok ? y : 0.f);
      }
    }
  e->n_drag = (int)ents.size();
  e->drag_T = T;
  e->cond_present_drag = true;   // DragPointEncoder emits its entry whenever the type has rows (:164-191)
  // (round 6, ADVICE round 5: through the engine's arena like every other setter -- outside a StageScope upload() fell to hipMemcpyAsync from
  // these pageable stack vectors on the engine's stream, safe only while the runtime keeps such copies host-synchronous, and behind an in-flight rollout)
  StageScope stage_scope(&e->stage[e->stage_cur], e->stream);
  if (e->n_drag > 0) {
    if (upload(e->d_drag_in, pts.data(), pts.size(), e->stream) || upload(e->d_drag_mask, pm.data(), pm.size(), e->stream) ||
        e->d_drag_emd.ensure((size_t)e->n_drag * D))
      return fail(PS_E_HIP, "drag-point upload failed");
  }
  e->ents_drag.swap(ents);
  const int rc = rebuild_conditions(e);   // (its mark() flushes the slices staged above with its own)
  if (rc == PS_OK) stage_scope.commit();
  return rc;
}

// Binary (agent-pair) tag conditions: 'v2v_tag' (condition_encoders.py:148-150; condition_attns.py:141-166).
extern "C" int ps_set_pair_conditions(ps_engine* e, int32_t C_pair, const float* pair_input, const uint8_t* pair_mask,
                                      const int32_t* pair_pidx) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_set_pair_conditions before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  e->ents_pair.clear();
  e->cond_present_pair = false;
  if (C_pair <= 0 || !pair_input) return rebuild_conditions(e);
  if (!pair_mask || !pair_pidx) return fail(PS_E_ARG, "pair_mask / pair_pidx missing");
  if (!e->cfg.v2v_tag_mask) return fail(PS_E_ARG, "this engine was created without binary tags (v2v_tag_mask = 0)");
  const int N = e->N;
  std::vector<int> slot2a((size_t)e->B * N, -1);
  for (int i = 0; i < e->Ap; ++i) slot2a[e->agent_rows[i]] = i;
  std::vector<ps_engine::CondEnt> ents;
  bool present = false;
  for (int b = 0; b < e->B; ++b)
    for (int c = 0; c < C_pair; ++c) {
      const size_t i = (size_t)b * C_pair + c;
      const int tag = (int)pair_input[3 * i];
      if (tag < 0 || tag > 4 || !(e->cfg.v2v_tag_mask & (1 << tag))) continue;   // not a used V2V tag value: no entry (:106-111)
      present = true;                                                           // the tag's key exists whatever the mask says
      if (!pair_mask[i]) continue;
      const int ns = pair_pidx[2 * i], nt = pair_pidx[2 * i + 1];
      if (ns < 0 || ns >= N || nt < 0 || nt >= N || ns == nt || slot2a[(size_t)b * N + ns] < 0 || slot2a[(size_t)b * N + nt] < 0)
        return fail(PS_E_ARG, "pair condition on an invalid pair of prompt slots");
      const int rs = slot2a[(size_t)b * N + ns], rt = slot2a[(size_t)b * N + nt];
      const float t0 = pair_input[3 * i + 1], t1 = pair_input[3 * i + 2];
      ents.push_back({rt, 3, 2 * tag, {0.f, t0, t1}, rs});       // edge s -> t carries the source half
      ents.push_back({rs, 3, 2 * tag + 1, {0.f, t0, t1}, rt});   // edge t -> s the target half
    }
  e->cond_present_pair = present;
  e->ents_pair.swap(ents);
  return rebuild_conditions(e);
}

extern "C" int ps_set_future_obs(ps_engine* e, const float* fut_input) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_set_future_obs before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq;
  const size_t per = (size_t)e->A * c.hist_steps * c.obs_dim;
  if (!fut_input || R < 2) { e->have_fut = false; return PS_OK; }
  std::vector<float> comp((size_t)(R - 1) * per);
  const size_t frame = (size_t)e->B * e->N * c.hist_steps * c.obs_dim, arow = (size_t)c.hist_steps * c.obs_dim;
  for (int r = 0; r < R - 1; ++r)
    for (int i = 0; i < e->A; ++i)
      std::memcpy(&comp[(size_t)r * per + i * arow], fut_input + r * frame + (size_t)e->agent_rows[i] * arow, sizeof(float) * arow);
  if (upload(e->d_fut, comp.data(), comp.size(), e->stream)) return fail(PS_E_HIP, "fut upload failed");
  HIPCHK(hipStreamSynchronize(e->stream));
  e->have_fut = true;
  drop_graph(e);
  return PS_OK;
}

extern "C" int ps_set_future_log(ps_engine* e, const float* fut_input, const uint8_t* fut_mask, const float* fut_pos,
                                 const float* fut_head) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_set_future_log before ps_set_scene");
  int rc = ps_set_future_obs(e, fut_input);
  if (rc) return rc;
  const ps_config& c = e->cfg;
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq;
  e->have_log = false;
  if (!fut_input || R < 2) return PS_OK;
  if (!fut_mask || !fut_pos || !fut_head) return fail(PS_E_ARG, "ps_set_future_log needs mask, position and heading frames");
  const int A = e->A;
  const size_t arow = (size_t)c.hist_steps * c.obs_dim, frame = (size_t)e->B * e->N;
  std::vector<uint8_t> mk((size_t)(R - 1) * A * arow);
  std::vector<float> ps((size_t)(R - 1) * A * 2), hd((size_t)(R - 1) * A);
  for (int r = 0; r < R - 1; ++r)
    for (int i = 0; i < A; ++i) {
      const size_t slot = (size_t)r * frame + e->agent_rows[i];
      std::memcpy(&mk[((size_t)r * A + i) * arow], fut_mask + slot * arow, arow);
      ps[((size_t)r * A + i) * 2] = fut_pos[slot * 2];
      ps[((size_t)r * A + i) * 2 + 1] = fut_pos[slot * 2 + 1];
      hd[(size_t)r * A + i] = fut_head[slot];
    }
  if (upload(e->d_fut_mask, mk.data(), mk.size(), e->stream) || upload(e->d_fut_pos, ps.data(), ps.size(), e->stream) ||
      upload(e->d_fut_head, hd.data(), hd.size(), e->stream))
    return fail(PS_E_HIP, "future log upload failed");
  HIPCHK(hipStreamSynchronize(e->stream));
  e->have_log = true;
  drop_graph(e);
  return PS_OK;
}

extern "C" int ps_set_replicas(ps_engine* e, int32_t replicas) {
  if (!e) return fail(PS_E_ARG, "null engine");
  if (replicas < 1) return fail(PS_E_ARG, "ps_set_replicas: need replicas >= 1");
  if (replicas != e->replicas_next) {   // takes effect with the next ps_set_scene; the current scene is gone
    e->replicas_next = replicas;
    e->have_scene = e->encoded = e->generated = e->reset = false;
    drop_graph(e);
  }
  return PS_OK;
}
extern "C" int32_t ps_num_replicas(ps_engine* e) { return e ? e->replicas : 0; }

extern "C" int ps_set_mode_choice(ps_engine* e, const int32_t* choice) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_set_mode_choice before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq, A = e->A;
  std::vector<int> rows((size_t)R * A, 0);
  if (choice)
    for (int r = 0; r < R; ++r)
      for (int i = 0; i < A; ++i) {
        const int k = choice[(size_t)r * std::max(e->B, e->replicas) * e->N + e->agent_slot[i]];   // [R][B or replicas][N]
        if (e->is_policy_h[i] && (k < 0 || k >= c.motion_k)) return fail(PS_E_ARG, "ps_set_mode_choice: mode index outside 0..motion_k-1");
        rows[(size_t)r * A + i] = e->is_policy_h[i] ? k : 0;
      }
  // (the captured graph reads the table through its device pointer: no re-capture)
  HIPCHK(hipMemcpyAsync(e->d_choice.p, rows.data(), sizeof(int) * rows.size(), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return PS_OK;
}

extern "C" int ps_set_action_noise(ps_engine* e, const float* noise) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_set_action_noise before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq, A = e->A;
  const bool was = e->have_noise;
  if (!noise) {
    e->have_noise = false;
    if (was) drop_graph(e);   // (the head launch carries the table pointer or NULL by value)
    return PS_OK;
  }
  const size_t per = (size_t)c.motion_k * c.target_steps * 2;
  std::vector<float> rows((size_t)R * A * per, 0.f);
  const size_t slots = (size_t)std::max(e->B, e->replicas) * e->N;
  for (int r = 0; r < R; ++r)
    for (int i = 0; i < A; ++i) {
      if (!e->is_policy_h[i]) continue;
      const float* src = noise + ((size_t)r * slots + e->agent_slot[i]) * per;
      std::copy(src, src + per, rows.begin() + ((size_t)r * A + i) * per);
    }
  const float* before = e->d_noise.p;
  if (e->d_noise.ensure(rows.size())) return fail(PS_E_HIP, "device allocation failed");
  HIPCHK(hipMemcpyAsync(e->d_noise.p, rows.data(), sizeof(float) * rows.size(), hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->have_noise = true;
  if (!was || before != e->d_noise.p) drop_graph(e);
  return PS_OK;
}

extern "C" int32_t ps_num_policy_agents(ps_engine* e) { return e ? e->n_policy : 0; }
// flags[i] = 1 if agent row i (the order of every per-agent result) is a policy agent, 0 if it replays the log
extern "C" int ps_policy_flags(ps_engine* e, int32_t* flags, int64_t capacity) {
  if (!e || !flags) return fail(PS_E_ARG, "null argument");
  if (capacity < e->A) return fail(PS_E_ARG, "destination too small");
  for (int i = 0; i < e->A; ++i) flags[i] = e->is_policy_h[i];
  return PS_OK;
}

// ------------------------------------------------------------------------------------------ launches
namespace {

// split-path exchange buffers, grown on demand (never inside a captured rollout: ps_set_scene sizes them first)
int io_for(ps_engine* e, int Nd, EdgeIO& io) {
  const size_t n = (size_t)std::max(Nd, 1);
  // a captured rollout holds these pointers by value: if a later caller (ps_policy_forward with more rows than
  // ps_set_scene sized the buffers for) makes any of them grow, the graph must be re-captured
  const float* before[9] = {e->io_q.p, e->io_qt.p, e->io_cq.p, e->io_ar.p, e->io_av.p, e->io_l.p, e->io_s.p, e->io_g.p, e->io_m.p};
  struct Regrow {
    ps_engine* e; const float* const* b;
    ~Regrow() {
      const float* after[9] = {e->io_q.p, e->io_qt.p, e->io_cq.p, e->io_ar.p, e->io_av.p, e->io_l.p, e->io_s.p, e->io_g.p, e->io_m.p};
      for (int i = 0; i < 9; ++i)
        if (b[i] && after[i] != b[i]) { drop_graph(e); break; }
    }
  } regrow{e, before};
  if (e->io_q.ensure(n * 128) || e->io_qt.ensure(n * 1024) || e->io_cq.ensure(n * 8) || e->io_ar.ensure(n * 1024) ||
      e->io_av.ensure(n * 128) || e->io_l.ensure(n * 8) || e->io_s.ensure(n * 128) || e->io_g.ensure(n * 128) || e->io_m.ensure(n * 8))
    return -1;
  io.q = e->io_q.p; io.qt = e->io_qt.p; io.cq = e->io_cq.p; io.ar = e->io_ar.p; io.av = e->io_av.p; io.l = e->io_l.p;
  io.s = e->io_s.p; io.g = e->io_g.p; io.m = e->io_m.p;
  return 0;
}

// One attention layer as three launches: k_node PRE (+ the rows' own k | v when kv_out is given: self-attention),
// k_edge_small (degree <= ES_MAXDEG), k_node POST.  `stp` is a device pointer to the layer's ChainStep.
// geo_edges: the edge half on k_edge16 (geometry records, ps_chain16.h) instead of k_edge_small (rel-PE operand images).
static bool xcd_on(int bit, bool dflt);
bool split_uses_rt(const ps_engine* e, int Nd, int kr) {   // the row-tile node halves (and with them the geometry-record edge kernel)
  return kr == 3 && !e->legacy_rows && ((Nd + 15) / 16 >= 256 || e->node_mt);
}
int launch_split_layer(ps_engine* e, float* x, int Nd, const ChainStep* stp, int kr, int maxdeg, float* kv_out, _Float16* khl_out, bool geo_edges = false) {
  EdgeIO io{};
  if (io_for(e, Nd, io)) return fail(PS_E_HIP, "split-path buffers");
  hipStream_t st = e->stream;
  const dim3 gn((Nd + ND_ROWS - 1) / ND_ROWS), ge((Nd + 3) / 4);
  const float eps = e->cfg.ln_eps;
  const ChainStep* none = nullptr;
  const int tiles = (Nd + 15) / 16;
  // (below 256 row tiles -- a single 1152-token scene has 72 -- the staged k_node with its 16-row workgroups fills the chip better:
  // 0.71 against 0.88 ms per scene encoding; from the 8-scene batch up the row-tile halves win, 1.53 against 1.73 ms)
  if (split_uses_rt(e, Nd, kr)) {
    // row-tile node halves (ps_rowtile.h): a wave owns 16 MT rows, no barriers.  MT: one tile per wave while the launch has fewer
    // waves than the chip has SIMDs (latency), more rows per weight fragment beyond that
    const int tiles = (Nd + 15) / 16;
    // (measured on the 9216-row s2s layers of the benchmark batch: alone on the GPU 1 tile per wave is fastest -- 1.55 / 1.76 / 1.99 ms
    // per scene encoding at 1 / 2 / 3 tiles; with four rollouts in flight 2 tiles give 22.7 M agent-steps/s against 22.5 / 22.5)
    const int mt = e->node_mt ? e->node_mt : ((e->chain_rows >= 8 && tiles >= 256) || tiles >= 4096 ? 2 : 1);
    const dim3 gr((unsigned)(((tiles + mt - 1) / mt + 3) / 4));
    if (mt == 1) hipLaunchKernelGGL(k_node_pre_rt<1>, gr, dim3(256), RT_LDS_BYTES, st, (const float*)x, Nd, stp, io, eps, kv_out, khl_out);
    else if (mt == 2) hipLaunchKernelGGL(k_node_pre_rt<2>, gr, dim3(256), RT_LDS_BYTES, st, (const float*)x, Nd, stp, io, eps, kv_out, khl_out);
#ifdef PS_EXPERIMENTS
    else hipLaunchKernelGGL(k_node_pre_rt<3>, gr, dim3(256), RT_LDS_BYTES, st, (const float*)x, Nd, stp, io, eps, kv_out, khl_out);
#else
    else return fail(PS_E_ARG, "3 row tiles per wave: experiments builds only");
#endif
    static const bool skip_edge = exp_env("PS_SKIP_S2S_EDGE") != nullptr;   // experiments only (timing; wrong results)
    if (skip_edge) {}
#ifdef PS_EXPERIMENTS
    else if (geo_edges && e->wg_edges) hipLaunchKernelGGL(k_edge16, dim3((unsigned)tiles), dim3(512), c16_edge_lds_bytes(), st, Nd, stp, io, (const float*)e->div32);
#endif
    else if (geo_edges) hipLaunchKernelGGL(k_edge_rows, dim3((unsigned)((Nd + ER_WAVES - 1) / ER_WAVES)), dim3(64 * ER_WAVES), ER_LDS_BYTES, st, Nd, stp, io, (const float*)e->div32, xcd_on(3, true) ? 1 : 0);
    else if (maxdeg <= 32) hipLaunchKernelGGL((k_edge_small<3, 2>), ge, dim3(256), es_lds_bytes<2>(), st, Nd, stp, io);
    else hipLaunchKernelGGL((k_edge_small<3, 8>), ge, dim3(256), es_lds_bytes<8>(), st, Nd, stp, io);
    if (mt == 1) hipLaunchKernelGGL(k_node_post_rt<1>, gr, dim3(256), RT_LDS_BYTES, st, x, Nd, stp, io, eps);
    else if (mt == 2) hipLaunchKernelGGL(k_node_post_rt<2>, gr, dim3(256), RT_LDS_BYTES, st, x, Nd, stp, io, eps);
#ifdef PS_EXPERIMENTS
    else hipLaunchKernelGGL(k_node_post_rt<3>, gr, dim3(256), RT_LDS_BYTES, st, x, Nd, stp, io, eps);
#endif
  } else if (kr == 3) {
    hipLaunchKernelGGL(k_node<3>, gn, dim3(256), ND_LDS_BYTES, st, x, Nd, none, stp, io, eps, kv_out, khl_out);
    if (maxdeg <= 32) hipLaunchKernelGGL((k_edge_small<3, 2>), ge, dim3(256), es_lds_bytes<2>(), st, Nd, stp, io);
    else hipLaunchKernelGGL((k_edge_small<3, 8>), ge, dim3(256), es_lds_bytes<8>(), st, Nd, stp, io);
    hipLaunchKernelGGL(k_node<3>, gn, dim3(256), ND_LDS_BYTES, st, x, Nd, stp, none, io, eps, (float*)nullptr, (_Float16*)nullptr);
  } else {
    hipLaunchKernelGGL(k_node<4>, gn, dim3(256), ND_LDS_BYTES, st, x, Nd, none, stp, io, eps, kv_out, khl_out);
    hipLaunchKernelGGL((k_edge_small<4, 8>), ge, dim3(256), es_lds_bytes<8>(), st, Nd, stp, io);
    hipLaunchKernelGGL(k_node<4>, gn, dim3(256), ND_LDS_BYTES, st, x, Nd, stp, none, io, eps, (float*)nullptr, (_Float16*)nullptr);
  }
  return hipGetLastError() == hipSuccess ? 0 : fail(PS_E_HIP, "split layer launch failed");
}

// which launches use the XCD-aware mapping (PS_XCD overrides for experiments: bit0 policy, bit1 a2a, bit2 generator, bit3 s2s)
static bool xcd_on(int bit, bool dflt) {
  static const int env = exp_env("PS_XCD") ? atoi(exp_env("PS_XCD")) : -1;
  return env < 0 ? dflt : ((env >> bit) & 1) != 0;
}
int launch_chain(ps_engine* e, float* x, int Nd, int step0, int nsteps, int maxdeg, bool timed = false,
                 const ChainStep* steps_override = nullptr, int force_T = 0, int kr_override = 0, const float* x_in = nullptr,
                 bool xcd = false, bool geo = false) {
  // geo: the steps' edge sets carry geometry records and NO operand images (use_geo1 below): the one-row build with k_chain16's edge body
  if (!x_in) x_in = x;   // in place unless the caller has the input rows elsewhere (saves a copy launch)
  // rel-PE width of the launch's steps: condition steps (and the test hook's arbitrary rows) use all 128 columns
  const int steps_host_kr = kr_override ? kr_override : (steps_override ? 3 : e->h_steps[step0].kr);
  const ChainStep* steps = steps_override ? steps_override : e->d_steps.p + step0;
  hipStream_t st = e->stream;
  // rows per workgroup (T).  More rows per workgroup share each weight load; the T >= 2 kernels are built for
  // two workgroups per CU (<= 256 registers, <= 78 KB LDS) so a co-resident workgroup hides latency: take the
  // largest T that still leaves >= 2 workgroups per CU.  Measured on the 1024-agent policy launch (end of round 1):
  // T=2 (512 WGs) 521 us, T=4 (256 WGs) 575 us, 4 rows on one 8-wave workgroup per CU (code 84) 563 us; earlier in
  // the round the one-workgroup-per-CU builds with register prefetch (BIG) at T=2 / T=4: 1005 / 879 us (they spill
  // even with 512 registers).
  // Below 512 rows (a single 128-agent scene: fewer workgroups than CUs) a row gets one 4-wave workgroup built for two
  // workgroups per CU (code 11: 256 registers, late weight prefetch): 274 us per 128-agent policy launch, against 280
  // for eight waves per row (code 18) and 292 for four waves with the whole register file and early prefetch (code 1)
  // -- and the workgroups of two pipelined rollouts co-reside on a CU.
  int T = Nd >= 2048 ? 4 : (Nd >= 512 ? 2 : 11);   // (11 = ONE row on a four-wave workgroup built for two workgroups per CU)
  static const int env_T = exp_env("PS_CHAIN_T") ? atoi(exp_env("PS_CHAIN_T")) : 0;   // experiments only
  if (env_T && Nd >= 512) T = env_T;
  if ((e->chain_rows == 2 || e->chain_rows == 4) && Nd >= 512) T = e->chain_rows;   // (1 / 8 / 16 address k_chain16 only)
  if (e->chain_rows >= 8 && Nd >= 512) T = 4;   // throughput mode of this kernel
  static const int env_TP = exp_env("PS_CHAIN_TP") ? atoi(exp_env("PS_CHAIN_TP")) : 0;   // experiments only: the policy launch alone
  if (env_TP && timed && Nd >= 512) T = env_TP;
  static const int env_T1 = exp_env("PS_CHAIN_T1") ? atoi(exp_env("PS_CHAIN_T1")) : 0;   // experiments only
  if (env_T1 && Nd < 512) T = env_T1;
  if (force_T) T = force_T;
  if (attn_lds_floats<1>(maxdeg) * sizeof(float) > 150 * 1024) return fail(PS_E_ARG, "degree bound too large for LDS");
  const float eps = e->cfg.ln_eps;
  static const int env_flags = exp_env("PS_CHAIN_FLAGS") ? atoi(exp_env("PS_CHAIN_FLAGS")) : 0;  // ablation only
  const int flags = env_flags | (xcd ? 1024 : 0);   // 1024: XCD-aware block -> row mapping (ps_device.h xcd_block)
  // phase clocks (tools/gpu_phase.py): PS_CHAIN_PROF=1 makes the timed policy launches accumulate cycles per phase
  static const bool want_prof = exp_env("PS_CHAIN_PROF") != nullptr;
  static unsigned long long* d_prof = nullptr;
  unsigned long long* prof = nullptr;
  if (want_prof && timed && e->time_chain) {
    if (!d_prof && hipMalloc(&d_prof, 16 * sizeof(unsigned long long)) != hipSuccess) d_prof = nullptr;
    if (d_prof) (void)hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), st);
    prof = d_prof;
  }
  if (timed && e->time_chain) (void)hipEventRecord(e->ev0, st);
  const size_t lds4 = attn_lds_floats<4>(maxdeg) * sizeof(float), lds2 = attn_lds_floats<2>(maxdeg) * sizeof(float),
               lds1 = attn_lds_floats<1>(maxdeg) * sizeof(float);
  const int kr = steps_host_kr;   // every step of a launch has the same rel-PE width
  // k_attn_chain builds in the library: 1 row (code 11: four waves, two workgroups per CU), 2 rows, 4 rows; rel-PE width 3 | 4;
  // the policy launch under its own symbol.  (Round 3 dropped the builds that were measured, parity-tested and never
  // selected: 1 row on 8 waves, 4 rows on 8 waves, 1 row with the whole register file.)
  if (T != 11 && T != 2 && T != 4) return fail(PS_E_ARG, "k_attn_chain takes 1 (code 11), 2 or 4 rows per workgroup");
  if (geo && (T != 11 || kr != 3)) return fail(PS_E_STATE, "the geometry-record edge phase of k_attn_chain is the one-row build's (no operand images were made for this launch)");
#define PS_LAUNCH(TT, KRR, POL, GRID, LDS) \
  hipLaunchKernelGGL((k_attn_chain<TT, 4, KRR, false, POL>), dim3(GRID), dim3(WG), LDS, st, x, x_in, Nd, steps, nsteps, maxdeg, eps, flags, prof, (const float*)e->div32)
  const bool pol = timed && kr == 3;
  if (T == 11 && geo && kr == 3) {   // one row per workgroup, edge phase on geometry records (k_chain16's edge body)
    const size_t ldsg = lds1 + G1_FLOATS * sizeof(float);
    if (pol) hipLaunchKernelGGL((k_attn_chain<1, 4, 3, false, true, true>), dim3(Nd), dim3(WG), ldsg, st, x, x_in, Nd, steps, nsteps, maxdeg, eps, flags, prof, (const float*)e->div32);
    else hipLaunchKernelGGL((k_attn_chain<1, 4, 3, false, false, true>), dim3(Nd), dim3(WG), ldsg, st, x, x_in, Nd, steps, nsteps, maxdeg, eps, flags, prof, (const float*)e->div32);
  } else if (T == 11) {
    if (pol) PS_LAUNCH(1, 3, true, Nd, lds1);
    else if (kr == 3) PS_LAUNCH(1, 3, false, Nd, lds1);
    else PS_LAUNCH(1, 4, false, Nd, lds1);
  } else if (T == 2) {
    if (pol) PS_LAUNCH(2, 3, true, (Nd + 1) / 2, lds2);
    else if (kr == 3) PS_LAUNCH(2, 3, false, (Nd + 1) / 2, lds2);
    else PS_LAUNCH(2, 4, false, (Nd + 1) / 2, lds2);
  } else {
    if (pol) PS_LAUNCH(4, 3, true, (Nd + 3) / 4, lds4);
    else if (kr == 3) PS_LAUNCH(4, 3, false, (Nd + 3) / 4, lds4);
    else PS_LAUNCH(4, 4, false, (Nd + 3) / 4, lds4);
  }
#undef PS_LAUNCH
  if (timed && e->time_chain) {
    (void)hipEventRecord(e->ev1, st);
    (void)hipEventSynchronize(e->ev1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e->ev0, e->ev1);
    if (prof) {
      unsigned long long h[16];
      (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
      const int Tr = T == 11 ? 1 : T, nwg = (Nd + Tr - 1) / Tr;
      double tot = 0;
      for (int i = 0; i < 16; ++i) tot += (double)h[i];
      fprintf(stderr, "[chain prof] T=%d wgs=%d %.1f us; mean cycles per workgroup per phase (share):", T, nwg, ms * 1e3);
      for (int i = 0; i < 16; ++i) fprintf(stderr, " %d:%.0f(%.1f%%)", i, (double)h[i] / nwg, 100.0 * h[i] / tot);
      fprintf(stderr, " total %.0f\n", tot / nwg);
    }
    e->chain_ms_sum += ms;
    e->chain_launches++;
  }
  return hipGetLastError() == hipSuccess ? 0 : fail(PS_E_HIP, "k_attn_chain launch failed");
}

void launch_kv(ps_engine* e, const float* x, int Ns, int layer0, int nlayers, float* kv, _Float16* khl, size_t layer_stride) {
  if (Ns <= 0 || nlayers <= 0) return;
  const unsigned gx = (Ns + PN_ROWS - 1) / PN_ROWS;
  hipLaunchKernelGGL(k_kv_proj, dim3(gx, nlayers, gx * nlayers <= 128 ? 2 : 1), dim3(WG), KV_LDS_BYTES, e->stream, x, Ns,
                     (const AttnW*)(e->d_layers + layer0), kv, khl, layer_stride, e->cfg.ln_eps);
}

// Row-tile PointNet (ps_rowtile.h): a polyline's P points take L lanes x MT row tiles (MT L >= P), a wave G = 16 / L polylines.
// (MT, L) by the row count: enough waves for the chip's 1024 SIMDs first, then the layout that wastes the fewest slots
// (P = 19: 5 tiles x 4 lanes = 20 slots; P = 11: 3 x 4 = 12 when there are many polylines, 1 x 16 when there are few).
// (force_mt: ps_engine::force_mt, the test hook of ps_test_pointnet_mt -- row tiles per wave, -1 = the staged kernel; per ENGINE since
// round 5: as a process global a test call on one engine changed the PointNet of every engine of the process)
struct RtShape { int mt, l; };
constexpr RtShape kRtShapes[] = {{1, 4}, {1, 8}, {1, 16}, {2, 8}, {2, 16}, {3, 4}, {3, 8}, {4, 8}, {5, 4}};   // the builds in the library: every P <= 32 fits (2, 16)
RtShape pointnet_shape(int n_rows, int P, bool throughput, int force_mt) {
  RtShape best{0, 0};
  double best_cost = 0;
  for (const RtShape& sh : kRtShapes) {
    if (force_mt >= 1 && sh.mt != force_mt) continue;
    if (sh.mt * sh.l < P) continue;
    const long waves = (n_rows + 16 / sh.l - 1) / (16 / sh.l);
    const long rounds = (waves + 1023) / 1024;   // one 4-wave workgroup per CU (128 KB of weight stages)
    // a wave's time ~ tiles + a fixed part (pooled-row GEMMs, stage fills).  Alone on the GPU the launch's latency counts: rounds x
    // that time; with several rollouts in flight (throughput mode) its CU time does: waves x that time -- 1024 agent histories are
    // 1024 one-tile waves on all 256 CUs alone (18 us), 256 three-tile waves on 64 CUs when the other CUs have other work
    const double cost = (throughput ? (double)waves : (double)rounds) * (sh.mt + 1.2);
    if (!best.mt || cost < best_cost) { best = sh; best_cost = cost; }
  }
  return best;
}
template <int MT, int L>
void launch_pointnet_rt(ps_engine* e, const PointNetW& w, const float* pts, const uint8_t* mask, const int* rows, int n_rows, int P,
                        int feat_mask_dim, float* out) {
  const int waves = (n_rows + 16 / L - 1) / (16 / L);
  hipLaunchKernelGGL((k_pointnet_rt<MT, L>), dim3((waves + 3) / 4), dim3(256), RT_LDS_BYTES, e->stream, w, pts, mask, rows, n_rows, P,
                     feat_mask_dim, out, e->cfg.ln_eps);
}
void launch_pointnet(ps_engine* e, const PointNetW& w, const float* pts, const uint8_t* mask, const int* rows, int n_rows,
                     int P, int feat_mask_dim, float* out) {
  if (n_rows <= 0) return;
  if (!e->legacy_rows && e->force_mt >= 0 && w.n_pre >= 1 && w.n_mid >= 1 && w.in_dim <= 32) {
    const RtShape sh = pointnet_shape(n_rows, P, e->chain_rows >= 8, e->force_mt);
#define PS_RT(MT_, L_) if (sh.mt == MT_ && sh.l == L_) { launch_pointnet_rt<MT_, L_>(e, w, pts, mask, rows, n_rows, P, feat_mask_dim, out); return; }
    PS_RT(1, 4) PS_RT(1, 8) PS_RT(1, 16) PS_RT(2, 8) PS_RT(2, 16) PS_RT(3, 4) PS_RT(3, 8) PS_RT(4, 8) PS_RT(5, 4)
#undef PS_RT
  }
  const int G = std::min(PN_G, PN_ROWS / P);   // polylines per workgroup (ps_set_scene bounds P <= 32)
  hipLaunchKernelGGL(k_pointnet_mfma, dim3((n_rows + G - 1) / G), dim3(256), PN_LDS_BYTES, e->stream, w, pts, mask, rows, n_rows, P,
                     feat_mask_dim, out, e->cfg.ln_eps);
}

// rel-PE of one or two edge sets as the MFMA operand images (device-side tile counts, grid-stride over tiles)
struct PeArgs {
  EdgeSet* es;
  const float *src_ori, *dst_pos, *dst_ori;
};
void launch_relpe(ps_engine* e, const PeArgs* a, int nsets) {
  PeSets ps{};
  size_t grid = 1;
  for (int i = 0; i < nsets; ++i) {
    EdgeSet& es = *a[i].es;
    ps.s[i] = PeSet{es.esrc.p, es.eoff.p, es.toff.p, es.tdst.p, es.nq, a[i].src_ori, a[i].dst_pos, a[i].dst_ori, es.rtA.p, es.rtT.p};
    grid = std::max(grid, std::min<size_t>(4096, es.cap_edges / 32 + (size_t)es.nq + 1));
  }
  hipLaunchKernelGGL(k_relpe_tiles, dim3((unsigned)grid, nsets), dim3(256), 0, e->stream, ps, (const float*)e->d_tok_pos.p, e->div32, e->cfg.ln_eps);
}
void launch_relpe(ps_engine* e, EdgeSet& es, const float* src_ori, const float* dst_pos, const float* dst_ori) {
  PeArgs a{&es, src_ori, dst_pos, dst_ori};
  launch_relpe(e, &a, 1);
}

// the per-edge geometry records of one or two edge sets (what k_chain16 rebuilds the rel-PE rows from)
void launch_geo(ps_engine* e, const PeArgs* a, int nsets, bool raw = false) {
  GeoSets gs{};
  size_t grid = 1;
  for (int i = 0; i < nsets; ++i) {
    EdgeSet& es = *a[i].es;
    gs.s[i] = GeoSet{es.esrc.p, es.edst.p, es.eoff.p, es.nq, a[i].src_ori, a[i].dst_pos, a[i].dst_ori, es.geo.p};
    grid = std::max(grid, std::min<size_t>(4096, es.cap_edges / GEO_THREADS + 1));
  }
  hipLaunchKernelGGL(k_edge_geo, dim3((unsigned)grid, nsets), dim3(GEO_THREADS), GEO_LDS_BYTES, e->stream, gs, (const float*)e->d_tok_pos.p, e->div32, e->cfg.ln_eps, raw ? 1 : 0);
}

// the learnable rel-PE rows of one edge set (after launch_geo made its geometry records): both operand images, KR = 4
void launch_pe_learn(ps_engine* e, EdgeSet& es, const PeLearnW& w) {
  const size_t pairs = (es.cap_edges / 32 + (size_t)es.nq + 2) / 2;
  hipLaunchKernelGGL(k_pe_learn, dim3((unsigned)std::min<size_t>(1024, std::max<size_t>(pairs, 1))), dim3(256), PL_LDS_BYTES, e->stream, w,
                     (const EdgeGeo*)es.geo.p, (const int*)es.eoff.p, (const int*)es.toff.p, (const int*)es.tdst.p, es.nq, es.rtA.p,
                     es.rtT.p, e->cfg.ln_eps);
}

// Fused chain, second generation (ps_chain16.h).  rows per workgroup: the engine's choice keeps >= 256 workgroups in a
// launch while it can (4 rows at 1024 destinations), ps_set_chain_rows overrides (16 = throughput mode).
// do the fused chains over Nd destination rows run on k_chain16?  (ps_set_chain_impl; 0 = in throughput mode, and in
// latency mode above 256 rows -- since round 6's node phases: an 8-scene rollout alone 5.7 ms, and 4.6 / 5.2 / 5.4 / 5.5 / 5.6 ms for 3 / 4 / 5 / 6 / 7
// scenes against 5.6 / 5.9 / 7.2 / 7.5 / 7.7 ms on k_attn_chain's 2- and 4-row builds (tools/gpu_r6_mid_batches.py); up to 256 rows the one-row
// k_attn_chain with k_chain16's edge body fills the chip better: 3.6 against 4.3 ms for two scenes)
// (part: 0 scene encoder, 1 generator, 2 policy -- a part with a LEARNABLE rel-PE cannot rebuild its rows from geometry
// records inside the kernel: it reads the operand images the edge-MLP kernel made, on k_attn_chain)
bool use_c16(const ps_engine* e, int Nd, int part) {
  if (e->pe_on[part]) return false;
  return e->chain_impl >= 2 || (e->chain_impl == 0 && (e->chain_rows >= 8 || (e->chain_rows == 0 && Nd > 256)));
}
// do the fused chains over Nd rows run on the ONE-ROW k_attn_chain with k_chain16's edge body (round 5)?  Where launch_chain picks one
// row per workgroup (below 512 rows: a single scene) in the default implementation; ps_set_chain_impl(1) keeps the operand-image edge
// phase (the cross-check path of the tests), PS_NO_GEO1: experiments.
bool use_geo1(const ps_engine* e, int Nd, int part) {
  static const bool off = exp_env("PS_NO_GEO1") != nullptr;
  static const bool t1 = exp_env("PS_CHAIN_T1") != nullptr;
  if (off || t1 || e->pe_on[part] || use_c16(e, Nd, part)) return false;
  // (up to 256 rows: the build holds 122 KB of LDS, one workgroup per CU -- 257 .. 511 rows would take two rounds of the chip where the
  // operand-image build co-locates two workgroups per CU)
  return e->chain_impl == 0 && Nd <= 256;
}
int chain16_rows(ps_engine* e, int Nd) {
  static const int env_rows = exp_env("PS_C16_ROWS") ? atoi(exp_env("PS_C16_ROWS")) : 0;   // experiments only
  // the fewest rows per workgroup that still put the launch on the chip in ONE round of at most 256 workgroups (round 6: a second round of small
  // workgroups costs more than rows twice as long -- 512 agents: 4.7 against 5.2 ms per rollout with 2 instead of 4 rows, tools/gpu_r6_mid_batches.py)
  int rows = Nd > 2048 ? 16 : (Nd > 1024 ? 8 : (Nd > 512 ? 4 : (Nd > 256 ? 2 : 1)));
  if (e->chain_rows >= 1 && e->chain_rows <= 16) rows = e->chain_rows;
  if (env_rows) rows = env_rows;
  return rows;
}
int launch_chain16(ps_engine* e, float* x, int Nd, const ChainStep* steps, int nsteps, bool timed, const float* x_in, bool xcd) {
  if (!x_in) x_in = x;
  const int nw = 8;   // eight waves per workgroup, one workgroup per CU (the two-workgroups-of-four build of round 2 was never selected: dropped)
  int rows = chain16_rows(e, Nd);
  {   // experiments: another tiling for the non-policy launches of the throughput mode (PS_C16_ROWS_SMALL: up to 1024 rows -- generator pairs, a2a; PS_C16_ROWS_S2S: beyond)
    static const int rs = exp_env("PS_C16_ROWS_SMALL") ? atoi(exp_env("PS_C16_ROWS_SMALL")) : 0, rb = exp_env("PS_C16_ROWS_S2S") ? atoi(exp_env("PS_C16_ROWS_S2S")) : 0;
    if (!timed && Nd <= 1024 && rs) rows = rs;
    if (!timed && Nd > 1024 && rb) rows = rb;
  }
  const int W = rows < nw ? nw / rows : 1;   // waves per row: each leaves its own partial sums (slot = part * Nd + row)
  const dim3 grid((Nd + rows - 1) / rows);   // (no exchange buffers: the phases of k_chain16 meet in LDS)
  hipStream_t st = e->stream;
  if (timed && e->time_chain) (void)hipEventRecord(e->ev0, st);
  static const bool want_prof = exp_env("PS_CHAIN_PROF") != nullptr;   // tools only: in-kernel phase clocks of the timed launches
  static unsigned long long* d_prof = nullptr;
  unsigned long long* prof = nullptr;
  if (want_prof && timed && e->time_chain) {
    if (!d_prof && hipMalloc(&d_prof, 48 * sizeof(unsigned long long)) != hipSuccess) d_prof = nullptr;
    if (d_prof) (void)hipMemsetAsync(d_prof, 0, 48 * sizeof(unsigned long long), st);
    prof = d_prof;
  }
#define PS_C16_(NWW, POL, ONE) \
  hipLaunchKernelGGL((k_chain16<NWW, POL, ONE>), grid, dim3(64 * NWW), c16_lds_bytes<NWW>(), st, x, x_in, Nd, rows, steps, nsteps, e->div32, e->cfg.ln_eps, xcd ? 1 : 0, prof)
#define PS_C16(NWW, POL) do { if (W == 1) PS_C16_(NWW, POL, true); else PS_C16_(NWW, POL, false); } while (0)
  if (timed) PS_C16(8, true); else PS_C16(8, false);
#undef PS_C16_
#undef PS_C16
  if (timed && e->time_chain) {
    (void)hipEventRecord(e->ev1, st);
    (void)hipEventSynchronize(e->ev1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e->ev0, e->ev1);
    if (prof) {
      unsigned long long h[48];
      (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
      const int nwg = (Nd + rows - 1) / rows;
      fprintf(stderr, "[chain16 prof] rows=%d nw=%d wgs=%d %.1f us; mean cycles per workgroup:", rows, nw, nwg, ms * 1e3);
      fprintf(stderr, " POST stages:");
      for (int i = 16; i < 32; ++i) fprintf(stderr, " %.0f", (double)h[i] / nwg);
      fprintf(stderr, " PRE stages:");
      for (int i = 32; i < 40; ++i) fprintf(stderr, " %.0f", (double)h[i] / nwg);
      fprintf(stderr, " |");
      static const char* nm[13] = {"PRE", "EDGE", "POST", "-", "rec+kissue", "fourier", "stage+score", "softmax", "a_r", "a_v", "row-epi", "row-pro", "tiles"};
      for (int i = 0; i < 13; ++i) fprintf(stderr, " %s:%.0f", nm[i], (double)h[i] / nwg);
      fprintf(stderr, "\n");
    }
    e->chain_ms_sum += ms;
    e->chain_launches++;
  }
  return hipGetLastError() == hipSuccess ? 0 : fail(PS_E_HIP, "k_chain16 launch failed");
}

// radius search + CSR + rel-PE for one or two edge sets over the same queries (count -> scan -> fill -> rel-PE: one
// launch each for all sets).  pe_mode: 1 = the operand images of k_attn_chain, 2 = the geometry records of k_chain16.
// up to this many queries a search with geometry records is ONE launch of a workgroup per query (k_radius_geo); beyond, the count / fill /
// record launches (measured: 128 queries 20.6 -> 11 us per search; at 1024 queries x 2 sets the one-launch form costs the pipelined headline 2 %)
struct RadArgs {
  EdgeSet* es;
  const int *r1, *r2;
  float r;
  int cap, self_base;
  const int* cand_ok = nullptr;   // optional candidate filter (RadSet::cand_ok)
  int cand_base = 0;
  const PeLearnW* learn = nullptr;   // the set's learnable rel-PE embedding, or nullptr (fixed Fourier rows per pe_mode)
};
void launch_radius(ps_engine* e, const RadArgs* a, int nsets, const float* qpos, const int* qscene, int nq, const float* src_ori,
                   const float* dst_ori, int pe_mode = 1, bool knn = false) {
  RadSets rs{};
  for (int i = 0; i < nsets; ++i) {
    EdgeSet& es = *a[i].es;
    rs.s[i] = RadSet{CandSet{e->d_tok_pos.p, a[i].r1, a[i].r2}, a[i].r * a[i].r, a[i].cap, a[i].self_base, es.cnt.p,
                     es.eoff.p, es.toff.p, es.tdst.p, es.esrc.p, es.edst.p, a[i].cand_ok, a[i].cand_base};
  }
  // a wave per query; from 1024 queries on 16 waves per workgroup instead of 4: the same waves on a quarter of the CUs -- these kernels wait for memory
  // (a chain of dependent loads per wave), and a CU that holds one of their waves cannot start a k_chain16 workgroup (248 registers x 2 waves per SIMD)
  const int wpb = nq >= 1024 ? 16 : 4, grid = (nq + wpb - 1) / wpb;
  hipStream_t st = e->stream;
  rs.scanned = nq > CSR_PREFIX_MAX_Q ? 1 : 0;
  {
    // round 5: search + CSR + geometry records in ONE launch (k_radius_geo, ps_chain16.h) where the records are all the rel-PE work of the
    // sets (fixed Fourier rows on the geometry-record chains) -- every search of a rollout in throughput mode and of a single scene
    bool learn_any = false;
    int capx = 1;
    bool flags_ok = true;
    for (int i = 0; i < nsets; ++i) {
      learn_any |= a[i].learn != nullptr;
      capx = std::max(capx, a[i].cap + 1);
      flags_ok = flags_ok && a[i].es->sync.p && a[i].es->sync.n >= 2 * (size_t)nq + 8;
    }
    if (e->search_impl != 1 && nq <= SEARCH_WG_MAX_Q && flags_ok && !knn && !learn_any && pe_mode == 2 && !rs.scanned && (size_t)capx * sizeof(int) <= 64 * 1024) {
      GeoSets gs{};
      RadSyncs sy{};
      for (int i = 0; i < nsets; ++i) {
        EdgeSet& es = *a[i].es;
        gs.s[i] = GeoSet{es.esrc.p, es.edst.p, es.eoff.p, es.nq, src_ori, qpos, dst_ori, es.geo.p};
        sy.flag[i] = reinterpret_cast<unsigned long long*>(es.sync.p);
      }
      hipLaunchKernelGGL(k_radius_geo, dim3(nq, nsets), dim3(64 * RG_WAVES), (size_t)capx * sizeof(int), st, rs, gs, sy, qpos, qscene, nq,
                         (const float*)e->d_tok_pos.p, (const float*)e->div32, e->cfg.ln_eps, e->search_impl == 2 ? 0 : 256);
      return;
    }
  }
  if (knn) {   // MODEL.REL_POS_EDGE_FUNC 'knn': the cap nearest instead of the first cap inside the radius (same CSR plumbing)
    hipLaunchKernelGGL(k_knn_sets<0>, dim3(grid, nsets), dim3(64 * wpb), 0, st, rs, qpos, qscene, nq);
    if (rs.scanned) hipLaunchKernelGGL(k_exclusive_scan, dim3(nsets), dim3(1024), 0, st, rs, nq);
    hipLaunchKernelGGL(k_knn_sets<1>, dim3(grid, nsets), dim3(64 * wpb), 0, st, rs, qpos, qscene, nq);
  } else {
    hipLaunchKernelGGL(k_radius<0>, dim3(grid, nsets), dim3(64 * wpb), 0, st, rs, qpos, qscene, nq);
    for (int i = 0; i < nsets; ++i)
      if (a[i].self_base >= 0)
        hipLaunchKernelGGL(k_radius_selfrank, dim3(grid), dim3(64 * wpb), 0, st, rs.s[i].cs, qpos, qscene, nq, rs.s[i].r2, a[i].cap,
                           a[i].self_base, a[i].es->cnt.p, a[i].cand_ok, a[i].cand_base);
    // (no scan launch up to CSR_PREFIX_MAX_Q queries: the fill pass computes its own prefix, csr_prefix)
    if (rs.scanned) hipLaunchKernelGGL(k_exclusive_scan, dim3(nsets), dim3(1024), 0, st, rs, nq);
    hipLaunchKernelGGL(k_radius<1>, dim3(grid, nsets), dim3(64 * wpb), 0, st, rs, qpos, qscene, nq);
  }
  PeArgs pe[2];
  bool learn = false;
  for (int i = 0; i < nsets; ++i) {
    pe[i] = PeArgs{a[i].es, src_ori, qpos, dst_ori};
    learn |= a[i].learn != nullptr;
  }
  if (learn) {   // (the two sets of a launch belong to the same part of the model: both learnable or neither)
    launch_geo(e, pe, nsets, true);
    for (int i = 0; i < nsets; ++i) launch_pe_learn(e, *a[i].es, *a[i].learn);
    return;
  }
  if (pe_mode & 1) launch_relpe(e, pe, nsets);
  if (pe_mode & 2) launch_geo(e, pe, nsets);
}
void launch_radius(ps_engine* e, EdgeSet& es, const int* r1, const int* r2, const float* qpos, const int* qscene, int nq, float r,
                   int cap, int self_base, const float* src_ori, const float* dst_ori, const int* cand_ok = nullptr,
                   int cand_base = 0, int pe_mode = 1, const PeLearnW* learn = nullptr, bool knn = false) {
  RadArgs a{&es, r1, r2, r, cap, self_base, cand_ok, cand_base, learn};
  launch_radius(e, &a, 1, qpos, qscene, nq, src_ori, dst_ori, pe_mode, knn);
}

}  // namespace

extern "C" int ps_encode_scene(ps_engine* e) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_encode_scene before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int Mv = e->Mv, A = e->A;
  const int Ap = e->Ap;   // rows the encoder computes (== A without replicas; the replicas' rows are copies, made at the end)
  hipStream_t st = e->stream;
  float* tok = e->d_tok.p;
  // agent token geometry back to the init poses (a previous rollout moved them)
  if (dev_copy2(st, e->d_tok_pos.p + 2 * (size_t)Mv, e->d_init_pos.p, sizeof(float) * 2 * A, e->d_tok_ori.p + Mv, e->d_init_head.p, sizeof(float) * A))
    return fail(PS_E_HIP, "device copy launch failed");
  launch_pointnet(e, e->pn_map, e->d_map_input.p, e->d_map_mask.p, e->d_map_rows.p, Mv, e->P, 0, tok);
  launch_pointnet(e, e->pn_obs, e->d_obs_input.p, e->d_obs_mask.p, e->d_agent_rows.p, Ap, c.hist_steps, c.obs_dim, tok + (size_t)Mv * D);
  // (experiments builds: PS_DBG_STOP = k ends the encoder after its k-th stage, read on EVERY call -- tools/gpu_stage_bisect.py digests what the stage left)
  const int dbg_stop = exp_env("PS_DBG_STOP") ? atoi(exp_env("PS_DBG_STOP")) : 0;
#define PS_DBG_STOP_AT(k) do { if (dbg_stop == (k)) { HIPCHK(hipGetLastError()); return PS_OK; } } while (0)
  PS_DBG_STOP_AT(1);
  // knn graphs (attn_fusion.py:107-109) + rel-PE (:111-112).  Agent rows that only enter the scene with a later fut_obs
  // frame are no tokens yet: not a candidate of any query (their own rows are computed and ignored).
  const int* live0 = e->have_dead0 ? (const int*)e->d_live0.p : nullptr;
  // the s2s layers as one-step k_chain16 launches (k | v projection + chain) instead of the split path: the default in THROUGHPUT mode
  // (ps_set_chain_rows >= 8; round 4: 23.4 against 22.6 M agent-steps/s on the benchmark workload with four rollouts in flight) and with
  // ps_set_chain_impl(3); the split path (k_node_*_rt / k_node + k_edge_small) is the default in latency mode, where it is the
  // faster one (one rollout alone on the GPU).  Round 3 kept this path behind a knob because another fp32 evaluation order of the
  // scene tokens re-rolled the workload's near-cut edges; with the scaled lo halves of round 4 (ps_device.h f16_los) every path holds
  // all 1024 agents of the workload within 1e-4 of the fp64 oracle (profiles/r04_parity.json).
  static const int env_s2s = exp_env("PS_S2S_C16") ? atoi(exp_env("PS_S2S_C16")) : -1;   // experiments only: 0 / 1 force
  const bool s2s_c16 = !e->pe_on[0] && use_c16(e, Mv + Ap, 0) &&
                       (env_s2s >= 0 ? env_s2s != 0 : (e->chain_impl == 3 || (e->chain_impl == 0 && e->chain_rows >= 8)));
  // the split path's edge half on geometry records (k_edge16) whenever its node halves are the row-tile kernels
  static const bool no_split0 = exp_env("PS_NO_SPLIT") != nullptr;
  const bool s2s_split_geo = !s2s_c16 && !no_split0 && e->e_s2s.maxdeg <= ES_MAXDEG && split_uses_rt(e, Mv + Ap, e->pe_on[0] ? 4 : 3);
  {
    CandSet ca{e->d_tok_pos.p, e->d_r_agent.p, nullptr};
    hipLaunchKernelGGL(k_knn, dim3((Ap + 3) / 4), dim3(256), 0, st, ca, (const float*)(e->d_tok_pos.p + 2 * (size_t)Mv),
                       (const int*)(e->d_tok_scene.p + Mv), Ap, c.agent_knn, (const int*)e->e_a2a.eoff.p, e->e_a2a.esrc.p, e->e_a2a.edst.p,
                       live0, Mv);
    CandSet csn{e->d_tok_pos.p, e->d_r_map.p, e->d_r_agent.p};
    hipLaunchKernelGGL(k_knn, dim3((Mv + Ap + 3) / 4), dim3(256), 0, st, csn, (const float*)e->d_tok_pos.p,
                       (const int*)e->d_tok_scene.p, Mv + Ap, c.scene_knn, (const int*)e->e_s2s.eoff.p, e->e_s2s.esrc.p, e->e_s2s.edst.p,
                       live0, Mv);
    // a2a edges index agents globally (Mv + i) for positions; kv rows are agent-local -> fixed up below
    const PeArgs pe[2] = {{&e->e_a2a, e->d_tok_ori.p, e->d_tok_pos.p + 2 * (size_t)Mv, e->d_tok_ori.p + Mv},
                          {&e->e_s2s, e->d_tok_ori.p, e->d_tok_pos.p, e->d_tok_ori.p}};
    if (e->pe_on[0]) {   // LEARNABLE_PE: geometry records, then the edge MLP writes the (128-column) operand images
      launch_geo(e, pe, 2, true);
      launch_pe_learn(e, e->e_a2a, e->pe_learn[0]);
      launch_pe_learn(e, e->e_s2s, e->pe_learn[1]);
    } else {   // per set: geometry records for the layers whose edge phase rebuilds its rows (k_chain16, k_edge16), operand images for the others
      const bool ga = use_c16(e, Ap, 0) || use_geo1(e, Ap, 0);
      const bool gs = s2s_c16 || (s2s_split_geo);
      if (ga && gs) launch_geo(e, pe, 2);
      else if (!ga && !gs) launch_relpe(e, pe, 2);
      else {
        if (ga) launch_geo(e, &pe[0], 1); else launch_relpe(e, &pe[0], 1);
        if (gs) launch_geo(e, &pe[1], 1); else launch_relpe(e, &pe[1], 1);
      }
    }
  }
  PS_DBG_STOP_AT(2);
  // 6 x (a2a on the agent rows in place, s2s on all rows)  (attn_fusion.py:117-119).  kv is indexed by
  // GLOBAL token row for both (the a2a projection fills rows Mv.. of the shared kv buffer).
  static const bool no_split = exp_env("PS_NO_SPLIT") != nullptr;   // experiments only
  // (every size: the split path is also the one that tracks the reference closest -- on a configs[2] scene whose fp64 rollout passes
  // 2.4e-6 rad from a +-pi cut it is the only s2s kernel that keeps every agent on the reference's side, tools/gpu_cut_paths.py --
  // and it is faster than the fused chain from one 1152-row scene up; PS_SPLIT_MIN: experiments)
  static const int split_min = exp_env("PS_SPLIT_MIN") ? atoi(exp_env("PS_SPLIT_MIN")) : 0;
  const bool split_s2s = !no_split && Mv + Ap >= split_min && e->e_s2s.maxdeg <= ES_MAXDEG;
  for (int i = 0; i < c.scene_layers; ++i) {
    launch_kv(e, tok + (size_t)Mv * D, Ap, e->L_a2a + i, 1, e->d_kv.p + (size_t)Mv * 256, e->d_kh.p + (size_t)Mv * 256, 0);
    PS_DBG_STOP_AT(3 + 3 * i);
    if (use_c16(e, Ap, 0)) {
      if (launch_chain16(e, tok + (size_t)Mv * D, Ap, e->d_steps.p + e->step_a2a + i, 1, false, nullptr, false)) return PS_E_HIP;
    } else if (launch_chain(e, tok + (size_t)Mv * D, Ap, e->step_a2a + i, 1, e->e_a2a.maxdeg, false, nullptr, 0, 0, nullptr, xcd_on(1, false), use_geo1(e, Ap, 0))) return PS_E_HIP;
    PS_DBG_STOP_AT(4 + 3 * i);
    if (s2s_c16) {
      launch_kv(e, tok, Mv + Ap, e->L_s2s + i, 1, e->d_kv.p, e->d_kh.p, 0);
      if (launch_chain16(e, tok, Mv + Ap, e->d_steps.p + e->step_s2s + i, 1, false, nullptr, xcd_on(3, true))) return PS_E_HIP;
    } else if (split_s2s) {
      // split layer: node work as 16-row MFMA GEMMs (k_node; its PRE half also makes the rows'
      // k | v, they are this self-attention layer's sources), the 32-neighbour edge phase one wave per token
      const ChainStep* stp = e->d_steps.p + e->step_s2s + i;
      if (launch_split_layer(e, tok, Mv + Ap, stp, e->pe_on[0] ? 4 : 3, e->e_s2s.maxdeg, e->d_kv.p, e->d_kh.p, s2s_split_geo)) return PS_E_HIP;
    } else {
      launch_kv(e, tok, Mv + Ap, e->L_s2s + i, 1, e->d_kv.p, e->d_kh.p, 0);
      if (launch_chain(e, tok, Mv + Ap, e->step_s2s + i, 1, e->e_s2s.maxdeg)) return PS_E_HIP;
    }
    PS_DBG_STOP_AT(5 + 3 * i);
  }
#undef PS_DBG_STOP_AT
  if (live0)   // rows outside the scene keep a zero token (what FUSION 'mlp' takes as the previous token when they enter)
    hipLaunchKernelGGL(k_zero_dead_rows, dim3((Ap * D + 255) / 256), dim3(256), 0, st, tok + (size_t)Mv * D, live0, Ap);
  if (e->replicas > 1)   // replica_batch_for_parallel_rollout (rollout/gpu_utils.py:59-123): every replica starts from the same tokens
    hipLaunchKernelGGL(k_fan_out_rows, dim3(std::min(2048, (A - Ap) * (D / 4) / 256 + 1)), dim3(256), 0, st, tok + (size_t)Mv * D, Ap, A, D);
  HIPCHK(hipGetLastError());
  e->encoded = true;
  e->generated = false;
  return PS_OK;
}

namespace {
// k_mlp_rows over n_rows rows (the last argument of every call): one row per workgroup up to 256 rows, eight beyond
void launch_mlp_rows(hipStream_t st, const Mlp3W& m, const float* in, const int* rows, int in_stride, float* out, int out_stride, float eps, int n_rows) {
  if (n_rows <= 256) hipLaunchKernelGGL(k_mlp_rows<1>, dim3((unsigned)n_rows), dim3(128), 0, st, m, in, rows, in_stride, out, out_stride, eps, n_rows);
  else hipLaunchKernelGGL(k_mlp_rows<8>, dim3((unsigned)((n_rows + 7) / 8)), dim3(128), 0, st, m, in, rows, in_stride, out, out_stride, eps, n_rows);
}
}  // namespace

extern "C" int ps_generate_policy(ps_engine* e) {
  if (!e || !e->encoded) return fail(PS_E_STATE, "ps_generate_policy before ps_encode_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int Mv = e->Mv, A = e->A;
  const int Ap = e->Ap;   // rows the generator computes (== A without replicas)
  hipStream_t st = e->stream;
  // prompt encoder (prompt_encoder/base.py:36-46)
  launch_mlp_rows(st, e->mlp_prompt, (const float*)e->d_prompt.p, (const int*)e->d_agent_rows.p,
                     c.prompt_dim, e->d_xp.p, D, c.ln_eps, Ap);
  // prompt poses (== the observed poses in the reference's batches; kept separate for generality)
  const float* ppos = e->d_prompt_pos.p;
  const float* pori = e->d_prompt_ori.p;
  const int* pscene = e->d_tok_scene.p + Mv;
  // p2p: radius_graph over prompts, loop=False (sym_coord.py:86); candidates = the scene's agents,
  // positions taken from the prompt poses -> stage them as the agent token geometry
  if (dev_copy2(st, e->d_tok_pos.p + 2 * (size_t)Mv, ppos, sizeof(float) * 2 * A, e->d_tok_ori.p + Mv, pori, sizeof(float) * A)) return fail(PS_E_HIP, "device copy launch failed");
  // (with log-replay agents in the scene only the policy agents are prompts: candidate filter)
  const int pe_gen = (use_c16(e, Ap, 1) || use_geo1(e, Ap, 1)) ? 2 : 1;   // k_chain16 rebuilds the rel-PE rows from geometry records, k_attn_chain streams operand images
  launch_radius(e, e->e_p2p, e->d_r_agent.p, nullptr, ppos, pscene, Ap, c.dec_prompt_radius, c.dec_max_neigh, Mv, e->d_tok_ori.p, pori,
                e->all_policy ? nullptr : (const int*)e->d_is_policy.p, Mv, pe_gen, pe_of(e, &e->e_p2p), c.rel_pos_knn != 0);
  // restore observed agent poses for the scene tokens, then s2p: radius over all scene tokens (:94)
  if (dev_copy2(st, e->d_tok_pos.p + 2 * (size_t)Mv, e->d_init_pos.p, sizeof(float) * 2 * A, e->d_tok_ori.p + Mv, e->d_init_head.p, sizeof(float) * A))
    return fail(PS_E_HIP, "device copy launch failed");
  launch_radius(e, e->e_s2p, e->d_r_map.p, e->d_r_agent.p, ppos, pscene, Ap, c.dec_scene_radius, c.dec_max_neigh, -1,
                e->d_tok_ori.p, pori, e->have_dead0 ? (const int*)e->d_live0.p : nullptr, Mv, pe_gen, pe_of(e, &e->e_s2p), c.rel_pos_knn != 0);
  // k|v of the (fixed) scene tokens for all s2p layers in one launch
  launch_kv(e, e->d_tok.p, Mv + Ap, e->L_s2p, c.dec_layers, e->d_kv_s2p.p, e->d_kh_s2p.p, (size_t)(Mv + Ap) * 256);
  const int md = std::max(e->e_p2p.maxdeg, e->e_s2p.maxdeg);
  for (int i = 0; i < c.dec_layers; ++i) {
    // p2p edges carry GLOBAL agent rows (Mv + j): project into rows Mv.. of the shared kv buffer
    launch_kv(e, e->d_xp.p, Ap, e->L_p2p + i, 1, e->d_kv.p + (size_t)Mv * 256, e->d_kh.p + (size_t)Mv * 256, 0);
    // (the last pair leaves its rows in d_emd, the policy embedding: rows in from d_xp, out to d_emd -- no copy launch)
    float* xo = i + 1 == c.dec_layers ? e->d_emd.p : e->d_xp.p;
    if (use_c16(e, Ap, 1)) {
      if (launch_chain16(e, xo, Ap, e->d_steps.p + e->step_dec + 2 * i, 2, false, e->d_xp.p, false)) return PS_E_HIP;
    } else if (launch_chain(e, xo, Ap, e->step_dec + 2 * i, 2, md, false, nullptr, 0, 0, e->d_xp.p, xcd_on(2, false), use_geo1(e, Ap, 1))) return PS_E_HIP;
  }
  if (c.dec_layers <= 0 && dev_copy(st, e->d_emd.p, e->d_xp.p, sizeof(float) * (size_t)Ap * D)) return fail(PS_E_HIP, "device copy launch failed");
  if (c.goal_pred_k > 0) {   // Decoder._goal_pred on the decoder's embedding (decoder/base.py:22-58, sym_coord.py:133-136)
    launch_mlp_rows(st, e->mlp_goal_prob, (const float*)e->d_emd.p, (const int*)nullptr, D,
                       e->d_goal_prob.p, c.goal_pred_k, c.ln_eps, Ap);
    launch_mlp_rows(st, e->mlp_goal_point, (const float*)e->d_emd.p, (const int*)nullptr, D,
                       e->d_goal_point.p, 2 * c.goal_pred_k, c.ln_eps, Ap);
  }
  // condition transformer at 'policy_decoder' (traj_sam.py:129-137)
  if (e->have_cond && c.cond_layers > 0) {
    if (e->n_drag > 0)
      launch_pointnet(e, e->pn_drag, e->d_drag_in.p, e->d_drag_mask.p, nullptr, e->n_drag, e->drag_T, 0, e->d_drag_emd.p);
    if (e->n_cond_edges > 0) {   // (a present type whose entries are all masked off: the layers still run, without edges)
      if (dev_zero(st, e->e_cnd.rtA.p, sizeof(_Float16) * (size_t)e->n_cond_tiles * 8192)) return fail(PS_E_HIP, "device fill launch failed");
      if (dev_zero(st, e->e_cnd.rtT.p, sizeof(_Float16) * (size_t)e->n_cond_tiles * 8192)) return fail(PS_E_HIP, "device fill launch failed");
      hipLaunchKernelGGL(k_cond_edges, dim3(e->n_cond_edges), dim3(128), 0, st, e->cond, (const int*)e->d_ent_off.p,
                         (const int*)e->d_ent_type.p, (const float*)e->d_ent_val.p, (const float*)e->d_drag_emd.p, e->n_cond_edges,
                         (const CondEdge*)e->d_cond_edges.p, ppos, pori, e->e_cnd.rtA.p, e->e_cnd.rtT.p, c.ln_eps);
    }
    if (dev_copy(st, e->d_xc.p, e->d_emd.p, sizeof(float) * (size_t)Ap * D)) return fail(PS_E_HIP, "device copy launch failed");
    for (int i = 0; i < c.cond_layers; ++i) {
      launch_kv(e, e->d_xc.p, Ap, e->L_cnd + i, 1, e->d_kv.p, e->d_kh.p, 0);
      if (launch_chain(e, e->d_xc.p, Ap, e->step_cnd + i, 1, e->e_cnd.maxdeg)) return PS_E_HIP;
    }
    hipLaunchKernelGGL(k_add_rows, dim3((Ap * D + 255) / 256), dim3(256), 0, st, e->d_emd.p, (const float*)e->d_xc.p, Ap * D);
  }
  // reconst_pred = pred_mlp(policy_emd) (act_decoder.py:133-135) -- constant over the replans
  if (!c.no_reconst_pred)   // (USE_GOAL_PRED_LOSS)
    launch_mlp_rows(st, e->mlp_pred, (const float*)e->d_emd.p, (const int*)nullptr, D,
                       e->d_reconst.p, 2, c.ln_eps, Ap);
  if (e->replicas > 1) {   // the replicas share the prompts' embeddings (gpu_utils.py:73-82): fan the Ap computed rows out
    auto fan = [&](float* p, int w) {
      hipLaunchKernelGGL(k_fan_out_rows, dim3(std::min(2048, (A - Ap) * std::max(w / 4, 1) / 256 + 1)), dim3(256), 0, st, p, Ap, A, w);
    };
    fan(e->d_emd.p, D);
    if (!c.no_reconst_pred) fan(e->d_reconst.p, 2);
    if (c.goal_pred_k > 0) { fan(e->d_goal_prob.p, c.goal_pred_k); fan(e->d_goal_point.p, 2 * c.goal_pred_k); }
  }
  // k|v of the map tokens for all m2p layers: map tokens never change during the rollout
  launch_kv(e, e->d_tok.p, Mv, e->L_m2p, c.pol_layers, e->d_kv_m2p.p, e->d_kh_m2p.p, (size_t)Mv * 256);
  if (c.obs_attn_update)   // ... and for the s2s layers that the per-replan observation update re-runs (map -> agents)
    launch_kv(e, e->d_tok.p, Mv, e->L_s2s, c.scene_layers, e->d_kv_um.p, e->d_kh_um.p, (size_t)Mv * 256);
  HIPCHK(hipGetLastError());
  e->generated = true;
  return PS_OK;
}

extern "C" int ps_reset_rollout(ps_engine* e) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_reset_rollout before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  hipStream_t st = e->stream;
  if (dev_copy2(st, e->d_traj.p, nullptr, sizeof(float) * (size_t)e->A * e->stride_steps * 4, e->d_vel.p, nullptr, sizeof(float) * (size_t)e->A * e->stride_steps * 2))
    return fail(PS_E_HIP, "device fill launch failed");
  const int n = e->A * c.hist_steps;
  hipLaunchKernelGGL(k_init_state, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)e->d_obs_input.p, (const int*)e->d_agent_rows.p,
                     e->A, c.hist_steps, c.obs_dim, e->stride_steps, e->d_traj.p, e->d_vel.p);
  HIPCHK(hipGetLastError());
  e->reset = true;
  e->env_ready = -1;
  return PS_OK;
}

extern "C" int ps_policy_step(ps_engine* e, int32_t t_idx) {
  if (!e || !e->generated || !e->reset) return fail(PS_E_STATE, "ps_policy_step needs generate_policy + reset_rollout");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq;
  if (t_idx < 0 || t_idx >= R) return fail(PS_E_ARG, "t_idx out of range");
  const int Mv = e->Mv, A = e->A;
  hipStream_t st = e->stream;
  const int last = c.hist_steps + t_idx * c.replan_freq;
  const float* stat = (e->have_fut && t_idx > 0) ? e->d_fut.p + (size_t)(t_idx - 1) * A * c.hist_steps * c.obs_dim : e->d_static_in.p;
  StepLog lg{};
  if (!e->all_policy) {   // log-replay agents: this replan's logged observation / validity / pose
    const size_t arow = (size_t)c.hist_steps * c.obs_dim;
    const bool fr = t_idx > 0 && e->have_fut, frl = t_idx > 0 && e->have_log;
    lg.is_policy = e->d_is_policy.p;
    lg.frame_in = fr ? e->d_fut.p + (size_t)(t_idx - 1) * A * arow : nullptr;
    lg.frame_mask = frl ? e->d_fut_mask.p + (size_t)(t_idx - 1) * A * arow : nullptr;
    lg.frame_pos = frl ? e->d_fut_pos.p + (size_t)(t_idx - 1) * A * 2 : nullptr;
    lg.frame_head = frl ? e->d_fut_head.p + (size_t)(t_idx - 1) * A : nullptr;
    lg.init_mask = e->d_obs_mask_rows.p;
    lg.obs_mask = e->d_obs_in_mask.p;
    lg.live = e->d_tok_live.p;
  }
  // step_env (traj_sam.py:205-274) -- unless the previous replan's head launch already ran it for this replan (k_policy_head_row's tail)
  const bool env_done = t_idx > 0 && e->env_ready == t_idx;
  e->env_ready = -1;
  if (!env_done)
  hipLaunchKernelGGL(k_step_env, dim3(A), dim3(64), 0, st, (const float*)e->d_traj.p, (const float*)e->d_vel.p, e->stride_steps, last,
                     c.hist_steps, c.dt, (const float*)e->d_init_pos.p, (const float*)e->d_init_head.p, stat, c.obs_dim, e->d_obs_in.p,
                     e->d_cur_pos.p, e->d_cur_ori.p, t_idx > 0 ? 1 : 0, t_idx > 0 ? e->d_tok_pos.p + 2 * (size_t)Mv : (float*)nullptr,
                     t_idx > 0 ? e->d_tok_ori.p + Mv : (float*)nullptr, lg, c.no_pred_vel ? 1 : 0);
  float* atok = e->d_tok.p + (size_t)Mv * D;
  if (t_idx > 0) {
    // update_scene_emb / _replace_old_obs (attn_fusion.py:205-250): re-encode agents, swap tokens + poses
    // (k_step_env already moved the agents' token poses)
    float* enc_out = c.obs_fusion_mlp ? e->d_obs_new.p : atok;
    if (e->all_policy)
      launch_pointnet(e, e->pn_obs, e->d_obs_in.p, (const uint8_t*)nullptr, (const int*)nullptr, A, c.hist_steps, -1, enc_out);
    else   // logged observations carry their own validity
      launch_pointnet(e, e->pn_obs, e->d_obs_in.p, (const uint8_t*)e->d_obs_in_mask.p, (const int*)nullptr, A, c.hist_steps, c.obs_dim, enc_out);
    const int* live = e->all_policy ? nullptr : (const int*)e->d_tok_live.p;
    if (c.obs_fusion_mlp)   // FUSION 'mlp' (attn_fusion.py:175-203): token = obs_update_mlp(cat(previous token, new observation))
      hipLaunchKernelGGL(k_obs_fuse, dim3(A), dim3(128), 0, st, e->mlp_obs_fuse, atok, (const float*)e->d_obs_new.p, live, c.ln_eps);
    if (c.obs_attn_update) {
      // ATTN_UPDATE (attn_fusion.py:136-173): the agents re-attend to each other (radius_graph, no self loops) and to
      // the map (radius) through the scene encoder's own a2a / s2s layers, at their new poses
      const float* qpos = e->d_tok_pos.p + 2 * (size_t)Mv;
      const float* qori = e->d_tok_ori.p + Mv;
      const RadArgs ru[2] = {{&e->e_ua, e->d_r_agent.p, nullptr, c.enc_agent_radius, c.scene_knn, Mv, live, Mv, pe_of(e, &e->e_ua)},
                             {&e->e_um, e->d_r_map.p, nullptr, c.enc_scene_radius, c.scene_knn, -1, nullptr, 0, pe_of(e, &e->e_um)}};
      launch_radius(e, ru, 2, qpos, e->d_tok_scene.p + Mv, A, e->d_tok_ori.p, qori);
      const int mdu = std::max(e->e_ua.maxdeg, e->e_um.maxdeg);
      for (int i = 0; i < c.scene_layers; ++i) {
        launch_kv(e, atok, A, e->L_a2a + i, 1, e->d_kv.p + (size_t)Mv * 256, e->d_kh.p + (size_t)Mv * 256, 0);
        if (launch_chain(e, atok, A, e->step_upd + 2 * i, 2, mdu)) return PS_E_HIP;
      }
      if (live && c.obs_fusion_mlp)   // agents outside the log are no tokens: their rows stay zero for the next fusion
        hipLaunchKernelGGL(k_zero_dead_rows, dim3((A * D + 255) / 256), dim3(256), 0, st, atok, live, A);
    }
  }
  // policy.forward (policy/base.py:19 -> temporal_ar.py:75 -> act_decoder.py:239-283)
  launch_kv(e, atok, A, e->L_a2p, c.pol_layers, e->d_kv_a2p.p, e->d_kh_a2p.p, (size_t)A * 256);
  const int* pscene = e->d_tok_scene.p + Mv;
  {
    // (a log-replay agent that dropped out of the log at this replan is no scene token: candidate filter)
    const RadArgs ra[2] = {{&e->e_a2p, e->d_r_agent.p, nullptr, c.pol_agent_radius, c.pol_max_neigh, -1,
                            e->all_policy ? nullptr : (const int*)e->d_tok_live.p, Mv, pe_of(e, &e->e_a2p)},
                           {&e->e_m2p, e->d_r_map.p, nullptr, c.pol_map_radius, c.pol_max_neigh, -1, nullptr, 0, pe_of(e, &e->e_m2p)}};
    launch_radius(e, ra, 2, e->d_cur_pos.p, pscene, A, e->d_tok_ori.p, e->d_cur_ori.p, (use_c16(e, A, 2) || use_geo1(e, A, 2)) ? 2 : 1, c.rel_pos_knn != 0);
  }
  const int md = std::max(e->e_a2p.maxdeg, e->e_m2p.maxdeg);
  if (e->policy_events && (int)e->pev.size() >= 2 * R) HIPCHK(hipEventRecord(e->pev[2 * t_idx], st));
  // the policy tokens enter every replan unchanged (d_emd); the fused features leave to d_fused
  if (use_c16(e, A, 2)) {
    if (launch_chain16(e, e->d_fused.p, A, e->d_steps.p + e->step_pol, 2 * c.pol_layers, true, e->d_emd.p, xcd_on(0, true))) return PS_E_HIP;
  } else if (launch_chain(e, e->d_fused.p, A, e->step_pol, 2 * c.pol_layers, md, true, nullptr, 0, 0, e->d_emd.p, xcd_on(0, true), use_geo1(e, A, 2))) return PS_E_HIP;
  if (e->policy_events && (int)e->pev.size() >= 2 * R) HIPCHK(hipEventRecord(e->pev[2 * t_idx + 1], st));
  {
    static const bool probe = exp_env("PS_POL_EDGE_PROBE") != nullptr;   // experiments only (timing: what the many-waves edge kernel needs for one a2p + m2p layer pair)
    if (probe && use_c16(e, A, 2)) {
      EdgeIO io{};
      if (io_for(e, A, io)) return PS_E_HIP;
      for (int k = 0; k < 2; ++k)
        hipLaunchKernelGGL(k_edge_rows, dim3((unsigned)((A + ER_WAVES - 1) / ER_WAVES)), dim3(64 * ER_WAVES), ER_LDS_BYTES, st, A, e->d_steps.p + e->step_pol + k, io, (const float*)e->div32, 1);
    }
  }
  // _compute_traj + step_agent_traj
  {
    float* mp_out = e->d_motion.p + (size_t)t_idx * A * c.motion_k * c.target_steps * c.state_dim;
    const float* nz = e->have_noise ? (const float*)(e->d_noise.p + (size_t)t_idx * A * c.motion_k * c.target_steps * 2) : (const float*)nullptr;
    const int vcol = c.no_pred_vel ? -1 : (c.pred_gmm ? 6 : 3);
    if (c.motion_k == 1 && !c.k_pred_mlp && !e->legacy_rows && A <= 128 && c.target_steps <= 256 && c.target_steps * c.state_dim <= 128) {
      // few agents (a single scene): one workgroup per agent, fp32 GEMVs with register-streamed weights -- the row-tile head below
      // would run on A / 64 workgroups and wait for its LDS stage fills (35 -> 17 us per replan at 128 agents: 4.41 -> 4.25 ms per
      // single-scene rollout).  By the agent count alone: the bits of a scene do not depend on the engine mode.  (Up to ONE scene of
      // 128 agents: at 256 rows the row-tile kernel has four workgroups and is as fast -- and configs[3] seed 0's two scenes hold a
      // cluster of near-cut agents that another fp32 summation order re-rolls, as the reference's own fp32 run does there:
      // tests/golden/ref_standins_demo_cfg3_seed0_b2.npz.)
      StepNext nx{};
      if (t_idx + 1 < R && e->search_impl != 1) {   // the next replan's step_env rides in this launch's tail (ps_set_search_impl(1): the launches of rounds 1-4)
        const int t1 = t_idx + 1;
        const size_t arow = (size_t)c.hist_steps * c.obs_dim;
        nx.on = 1;
        nx.last = c.hist_steps + t1 * c.replan_freq;
        nx.hist = c.hist_steps; nx.obs_dim = c.obs_dim; nx.fd_vel = c.no_pred_vel ? 1 : 0; nx.dt = c.dt;
        nx.init_pos = e->d_init_pos.p; nx.init_head = e->d_init_head.p;
        nx.static_in = e->have_fut ? e->d_fut.p + (size_t)(t1 - 1) * A * arow : e->d_static_in.p;
        nx.obs_in = e->d_obs_in.p; nx.cur_pos = e->d_cur_pos.p; nx.cur_ori = e->d_cur_ori.p;
        nx.tok_pos = e->d_tok_pos.p + 2 * (size_t)Mv; nx.tok_ori = e->d_tok_ori.p + Mv;
        if (!e->all_policy) {
          nx.lg.is_policy = e->d_is_policy.p;
          nx.lg.frame_in = e->have_fut ? e->d_fut.p + (size_t)(t1 - 1) * A * arow : nullptr;
          nx.lg.frame_mask = e->have_log ? e->d_fut_mask.p + (size_t)(t1 - 1) * A * arow : nullptr;
          nx.lg.frame_pos = e->have_log ? e->d_fut_pos.p + (size_t)(t1 - 1) * A * 2 : nullptr;
          nx.lg.frame_head = e->have_log ? e->d_fut_head.p + (size_t)(t1 - 1) * A : nullptr;
          nx.lg.init_mask = e->d_obs_mask_rows.p;
          nx.lg.obs_mask = e->d_obs_in_mask.p;
          nx.lg.live = e->d_tok_live.p;
        }
        e->env_ready = t1;
      }
      hipLaunchKernelGGL(k_policy_head_row, dim3(A), dim3(256), 0, st, e->head, (const float*)e->d_fused.p, (const int*)e->d_agent_type.p, A,
                         c.target_steps, c.state_dim, mp_out, e->d_traj.p, e->d_vel.p, e->stride_steps, last, c.replan_freq, c.ln_eps, nz, vcol, nx);
    } else if (c.motion_k == 1 && !c.k_pred_mlp && !e->legacy_rows) {
      // row-tile head (ps_rowtile.h): a wave carries 16 agents through CG_decode and the motion head in registers
      hipLaunchKernelGGL(k_policy_head_rt, dim3((A + 63) / 64), dim3(256), RT_LDS_BYTES, st, e->head, (const float*)e->d_fused.p,
                         (const int*)e->d_agent_type.p, A, c.target_steps, c.state_dim, mp_out, e->d_traj.p, e->d_vel.p, e->stride_steps, last,
                         c.replan_freq, c.ln_eps, nz, vcol);
    } else {
      const int G = c.k_pred_mlp ? 16 : 16 / c.motion_k;   // agents per workgroup: one 16-row tile holds G agents x K modes (PRED_MODE mlp: 16 agents)
      hipLaunchKernelGGL(k_policy_head_mfma, dim3((A + G - 1) / G), dim3(256), 0, st, e->head, (const float*)e->d_fused.p,
                         (const int*)e->d_agent_type.p, A, c.motion_k, c.target_steps, c.state_dim, mp_out,
                         e->d_traj.p, e->d_vel.p, e->stride_steps, last, c.replan_freq, c.ln_eps, (const int*)(e->d_choice.p + (size_t)t_idx * A),
                         nz, vcol, c.k_pred_mlp ? 1 : 0);
    }
  }
  HIPCHK(hipGetLastError());
  return PS_OK;
}

namespace {
int rollout_eager(ps_engine* e) {
  int rc;
  if ((rc = ps_encode_scene(e))) return rc;
  if ((rc = ps_generate_policy(e))) return rc;
  if ((rc = ps_reset_rollout(e))) return rc;
  const int R = (e->cfg.max_steps + e->cfg.replan_freq - 1) / e->cfg.replan_freq;
  for (int t = 0; t < R; ++t)
    if ((rc = ps_policy_step(e, t))) return rc;
  return PS_OK;
}
// Round 5: a setter no longer DESTROYS the captured rollout, it only marks it unchecked.  The next ps_rollout compares the signature
// of what its launch sequence would be made of NOW -- every count, flag and device pointer a launch of rollout_eager() takes from
// the host -- with the signature taken at capture; equal means the recorded launches are the ones a new capture would record (the
// DATA behind the pointers is read by the kernels, grids that depend on data read their counts on the device), so a stream of
// batches of one shape replays one graph.  Anything that moved (a buffer that grew, another row count, a condition type that
// appeared) re-captures as before.
void drop_graph(ps_engine* e) { e->graph_ok = false; e->env_ready = -1; }   // (every setter comes through here: whatever a fused step_env read may have changed)
void destroy_graph(ps_engine* e) {
  if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
  if (e->graph) (void)hipGraphDestroy(e->graph);
  e->graph_exec = nullptr;
  e->graph = nullptr;
  e->graph_ok = false;
  e->graph_sig.clear();
}
std::vector<uint64_t> rollout_signature(const ps_engine* e) {
  std::vector<uint64_t> g;
  g.reserve(256);
  auto I = [&](long long v) { g.push_back((uint64_t)v); };
  auto Pp = [&](const void* p) { g.push_back((uint64_t)(uintptr_t)p); };
  // shapes and row counts
  for (long long v : {(long long)e->B, (long long)e->M, (long long)e->P, (long long)e->N, (long long)e->Mv, (long long)e->A, (long long)e->Ap,
                      (long long)e->replicas, (long long)e->maxA_scene, (long long)e->maxM_scene, (long long)e->n_policy, (long long)e->stride_steps})
    I(v);
  // flags that choose launches, pointers or NULLs
  for (long long v : {(long long)e->all_policy, (long long)e->have_log, (long long)e->have_dead0, (long long)e->have_fut, (long long)e->have_noise,
                      (long long)e->have_cond, (long long)e->n_cond_edges, (long long)e->n_cond_tiles, (long long)e->n_drag, (long long)e->drag_T,
                      (long long)e->node_mt, (long long)e->wg_edges, (long long)e->legacy_rows, (long long)e->chain_rows, (long long)e->chain_impl,
                      (long long)e->force_mt, (long long)e->search_impl, (long long)e->step_a2a, (long long)e->step_s2s, (long long)e->step_dec, (long long)e->step_cnd,
                      (long long)e->step_pol, (long long)e->step_upd})
    I(v);
  I((long long)e->h_steps.size());
  for (const ChainStep& st : e->h_steps) I(st.kr);
  // every device buffer a launch can name
  for (const void* p : {(const void*)e->io_q.p, (const void*)e->io_qt.p, (const void*)e->io_cq.p, (const void*)e->io_ar.p, (const void*)e->io_av.p,
                        (const void*)e->io_l.p, (const void*)e->io_s.p, (const void*)e->io_g.p, (const void*)e->io_m.p,
                        (const void*)e->d_obs_new.p, (const void*)e->d_kv_um.p, (const void*)e->d_kh_um.p,
                        (const void*)e->d_map_input.p, (const void*)e->d_obs_input.p, (const void*)e->d_prompt.p, (const void*)e->d_fut.p,
                        (const void*)e->d_map_mask.p, (const void*)e->d_obs_mask.p, (const void*)e->d_map_rows.p, (const void*)e->d_agent_rows.p,
                        (const void*)e->d_tok_scene.p, (const void*)e->d_agent_type.p, (const void*)e->d_r_map.p, (const void*)e->d_r_agent.p,
                        (const void*)e->d_r_zero.p, (const void*)e->d_tok.p, (const void*)e->d_tok_pos.p, (const void*)e->d_tok_ori.p,
                        (const void*)e->d_init_pos.p, (const void*)e->d_init_head.p, (const void*)e->d_cur_pos.p, (const void*)e->d_cur_ori.p,
                        (const void*)e->d_prompt_pos.p, (const void*)e->d_prompt_ori.p, (const void*)e->d_xp.p, (const void*)e->d_emd.p,
                        (const void*)e->d_xc.p, (const void*)e->d_fused.p, (const void*)e->d_obs_in.p, (const void*)e->d_static_in.p,
                        (const void*)e->d_kv.p, (const void*)e->d_kv_s2p.p, (const void*)e->d_kv_m2p.p, (const void*)e->d_kv_a2p.p,
                        (const void*)e->d_traj.p, (const void*)e->d_vel.p, (const void*)e->d_motion.p, (const void*)e->d_reconst.p,
                        (const void*)e->d_goal_prob.p, (const void*)e->d_goal_point.p, (const void*)e->d_choice.p, (const void*)e->d_noise.p,
                        (const void*)e->d_kh.p, (const void*)e->d_kh_s2p.p, (const void*)e->d_kh_m2p.p, (const void*)e->d_kh_a2p.p,
                        (const void*)e->d_steps.p, (const void*)e->d_ent_off.p, (const void*)e->d_ent_type.p, (const void*)e->d_ent_val.p,
                        (const void*)e->d_cond_edges.p, (const void*)e->d_drag_in.p, (const void*)e->d_drag_emd.p, (const void*)e->d_drag_mask.p,
                        (const void*)e->d_is_policy.p, (const void*)e->d_tok_live.p, (const void*)e->d_live0.p, (const void*)e->d_obs_in_mask.p,
                        (const void*)e->d_fut_mask.p, (const void*)e->d_obs_mask_rows.p, (const void*)e->d_fut_pos.p, (const void*)e->d_fut_head.p})
    Pp(p);
  for (const EdgeSet* es : {&e->e_a2a, &e->e_s2s, &e->e_p2p, &e->e_s2p, &e->e_a2p, &e->e_m2p, &e->e_cnd, &e->e_ua, &e->e_um}) {
    Pp(es->cnt.p); Pp(es->eoff.p); Pp(es->toff.p); Pp(es->tdst.p); Pp(es->esrc.p); Pp(es->edst.p); Pp(es->rtA.p); Pp(es->rtT.p); Pp(es->geo.p); Pp(es->sync.p);
    I((long long)es->cap_edges); I(es->nq); I(es->maxdeg);
  }
  return g;
}
}  // namespace

// The ~330 launches of one rollout (everything is enqueued on one stream, nothing syncs with the
// host) are captured into a hipGraph the first time and replayed afterwards: the launch-bound tail
// of small kernels (radius / scan / rel-PE / kv projections) stops paying per-launch host time.
extern "C" int ps_rollout(ps_engine* e) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_rollout before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  // (event pairs do not survive graph capture on ROCm 7.2 -- hipEventElapsedTime rejects events recorded by graph nodes --
  // so a rollout with policy events enabled is launched eagerly: ~200 launches, ~1 ms of host time per rollout)
  if (!e->use_graph || e->time_chain || e->policy_events) return rollout_eager(e);
  if (!e->graph_ok && e->graph_exec) {   // a setter ran since the last replay: does the recorded launch sequence still stand?
    if (rollout_signature(e) == e->graph_sig) {
      e->graph_ok = true;
      e->graph_reuses++;
    }
  }
  if (!e->graph_ok) {
    destroy_graph(e);
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    const int rc = rollout_eager(e);
    hipGraph_t g = nullptr;
    const hipError_t ce = hipStreamEndCapture(e->stream, &g);
    if (rc || ce != hipSuccess || !g) {
      if (g) (void)hipGraphDestroy(g);
      e->use_graph = false;   // fall back to eager launches (still the HIP path)
      (void)hipGetLastError();
      return rollout_eager(e);
    }
    e->graph = g;
    if (hipGraphInstantiate(&e->graph_exec, g, nullptr, nullptr, 0) != hipSuccess) {
      destroy_graph(e);
      e->use_graph = false;
      (void)hipGetLastError();
      return rollout_eager(e);
    }
    e->graph_ok = true;
    e->graph_sig = rollout_signature(e);   // (after the capture: rollout_eager may have grown an exchange buffer)
    e->graph_captures++;
  }
  HIPCHK(hipGraphLaunch(e->graph_exec, e->stream));
  e->encoded = e->generated = e->reset = true;
  return PS_OK;
}

extern "C" int64_t ps_graph_nodes(ps_engine* e) {
  if (!e || !e->graph_ok || !e->graph) return 0;
  size_t n = 0;
  if (hipGraphGetNodes(e->graph, nullptr, &n) != hipSuccess) return -1;
  return (int64_t)n;
}

extern "C" int ps_sync(ps_engine* e) {
  if (!e) return fail(PS_E_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream));
  return PS_OK;
}

// scene_encoder.update_scene_emb (attn_fusion.py:238-252 with OBS_UPDATE {FUSION: replace, ATTN_UPDATE: False}): re-encode
// the agents from a new observation, replace their tokens and poses, keep the map tokens.
extern "C" int ps_update_obs(ps_engine* e, const float* obs_input, const uint8_t* obs_mask, const float* obs_pos,
                             const float* obs_head) {
  if (!e || !e->have_scene || !e->encoded) return fail(PS_E_STATE, "ps_update_obs before ps_encode_scene");
  if (!obs_input || !obs_mask || !obs_pos || !obs_head) return fail(PS_E_ARG, "ps_update_obs: null argument");
  if (e->cfg.obs_fusion_mlp || e->cfg.obs_attn_update)
    return fail(PS_E_ARG, "ps_update_obs implements OBS_UPDATE {FUSION: replace, ATTN_UPDATE: False}; the variants run inside ps_policy_step");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int A = e->A, Mv = e->Mv;
  const size_t arow = (size_t)c.hist_steps * c.obs_dim;
  std::vector<float> in((size_t)A * arow), pos((size_t)A * 2), ori(A);
  std::vector<uint8_t> mk((size_t)A * arow);
  for (int i = 0; i < A; ++i) {
    const size_t r = e->agent_rows[i];
    bool any = false;
    for (int h = 0; h < c.hist_steps && !any; ++h) {
      bool all = true;
      for (int f = 0; f < c.obs_dim; ++f) all &= obs_mask[r * arow + (size_t)h * c.obs_dim + f] != 0;
      any = all;
    }
    if (!any) return fail(PS_E_ARG, "ps_update_obs: the observed agents must be those of ps_set_scene (an agent lost every history step)");
    for (size_t k = 0; k < arow; ++k) {
      mk[i * arow + k] = obs_mask[r * arow + k];
      const float v = obs_input[r * arow + k];
      in[i * arow + k] = std::isnan(v) ? 0.f : v;
    }
    pos[2 * i] = obs_pos[2 * r];
    pos[2 * i + 1] = obs_pos[2 * r + 1];
    ori[i] = obs_head[r];
  }
  DevBuf<float> d_in, d_pos, d_ori;
  DevBuf<uint8_t> d_mk;
  hipStream_t st = e->stream;
  if (upload(d_in, in.data(), in.size(), st) || upload(d_mk, mk.data(), mk.size(), st) || upload(d_pos, pos.data(), pos.size(), st) ||
      upload(d_ori, ori.data(), ori.size(), st))
    return fail(PS_E_HIP, "ps_update_obs upload failed");
  launch_pointnet(e, e->pn_obs, d_in.p, d_mk.p, nullptr, A, c.hist_steps, c.obs_dim, e->d_tok.p + (size_t)Mv * D);
  if (dev_copy(st, e->d_tok_pos.p + 2 * (size_t)Mv, d_pos.p, sizeof(float) * 2 * A)) return fail(PS_E_HIP, "device copy launch failed");
  if (dev_copy(st, e->d_tok_ori.p + Mv, d_ori.p, sizeof(float) * A)) return fail(PS_E_HIP, "device copy launch failed");
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(st));
  d_in.release(); d_mk.release(); d_pos.release(); d_ori.release();
  e->generated = false;   // a policy generated from the old tokens is stale
  return PS_OK;
}

// The map tokens of an encoded scene, handed back after a ps_set_scene with the SAME map and ANOTHER agent set:
// update_scene_emb keeps the map part of scene_embs and takes whatever agents the new observation lists
// (_replace_old_obs, attn_fusion.py:205-236).  Marks the scene encoded; the agent tokens are whatever the next
// ps_update_obs makes them.
extern "C" int ps_set_map_tokens(ps_engine* e, const float* tokens, int64_t count) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_set_map_tokens before ps_set_scene");
  if (!tokens || count != (int64_t)e->Mv * D) return fail(PS_E_ARG, "ps_set_map_tokens: expected [map tokens, hidden] floats");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipMemcpyAsync(e->d_tok.p, tokens, sizeof(float) * (size_t)count, hipMemcpyHostToDevice, e->stream));
  if (dev_zero(e->stream, e->d_tok.p + (size_t)e->Mv * D, sizeof(float) * (size_t)e->A * D)) return fail(PS_E_HIP, "device fill launch failed");
  HIPCHK(hipStreamSynchronize(e->stream));
  e->encoded = true;
  e->generated = false;
  drop_graph(e);
  return PS_OK;
}

extern "C" int ps_set_chain_rows(ps_engine* e, int32_t rows) {
  if (!e) return fail(PS_E_ARG, "null engine");
  if (rows < 0 || rows > 16 || (rows < 8 && rows != 0 && rows != 1 && rows != 2 && rows != 4)) return fail(PS_E_ARG, "ps_set_chain_rows: 0 (auto), 1, 2, 4, or 8..16");
  if (e->chain_impl == 1 && rows != 0 && rows != 2 && rows != 4) return fail(PS_E_ARG, "ps_set_chain_rows: k_attn_chain (ps_set_chain_impl 1) takes 0, 2 or 4");
  if (rows != e->chain_rows) drop_graph(e);
  e->chain_rows = rows;
  return PS_OK;
}

extern "C" int ps_set_chain_impl(ps_engine* e, int32_t impl) {
  if (!e) return fail(PS_E_ARG, "null engine");
  if (impl < 0 || impl > 3) return fail(PS_E_ARG, "ps_set_chain_impl: 0 (by mode), 1 (k_attn_chain), 2 (k_chain16) or 3 (k_chain16, the encoder's s2s layers too)");
  if (impl != e->chain_impl) {
    drop_graph(e);
    e->chain_rows = 0;
    e->encoded = e->generated = false;   // the edge sets of the finished stages carry the other kernel's rel-PE form
  }
  e->chain_impl = impl;
  return PS_OK;
}

extern "C" int ps_set_row_impl(ps_engine* e, int32_t impl) {
  if (!e) return fail(PS_E_ARG, "null engine");
  if (impl != 0 && impl != 1 && impl != 2 && !(impl >= 11 && impl <= 13))
    return fail(PS_E_ARG, "ps_set_row_impl: 0 = row-tile kernels (default), 1 = the round-3 staged kernels, 2 = row-tile kernels with the workgroup edge kernel, 11..13 = row-tile kernels with 1..3 row tiles per wave in the node halves");
#ifndef PS_EXPERIMENTS
  // round 5: the kernels behind 2 (k_edge16) and 13 (the 3-tile node halves: 292 - 656 B of scratch per lane) exist
  // for cross-checks only and are compiled into experiments builds (hipcc -DPS_EXPERIMENTS), not into the product library
  if (impl == 2 || impl == 13) return fail(PS_E_ARG, "ps_set_row_impl(2 | 13): experiments build only (hipcc -DPS_EXPERIMENTS, tools/README.md)");
#endif
  drop_graph(e);
  e->encoded = e->generated = false;   // (as ps_set_chain_impl: the s2s path of the encoder depends on it)
  e->legacy_rows = impl == 1;
  e->wg_edges = impl == 2;
  e->node_mt = impl >= 11 ? impl - 10 : 0;
  return PS_OK;
}

extern "C" int ps_set_search_impl(ps_engine* e, int32_t impl) {
  if (!e) return fail(PS_E_ARG, "null engine");
  if (impl != 0 && impl != 1 && impl != 2)
    return fail(PS_E_ARG, "ps_set_search_impl: 0 = one launch per radius search with geometry records (default), 1 = count / fill / record launches, 2 = as 0 with a look-back that never waits (tests)");
  drop_graph(e);
  e->search_impl = impl;
  return PS_OK;
}

extern "C" int ps_enable_policy_events(ps_engine* e, int32_t on) {
  if (!e) return fail(PS_E_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  const int R = (e->cfg.max_steps + e->cfg.replan_freq - 1) / e->cfg.replan_freq;
  if (on && e->pev.empty()) {
    e->pev.resize(2 * R, nullptr);
    for (auto& ev : e->pev) HIPCHK(hipEventCreate(&ev));
  }
  if ((on != 0) != e->policy_events) drop_graph(e);
  e->policy_events = on != 0;
  return PS_OK;
}
extern "C" int ps_policy_event_times(ps_engine* e, float* ms, int32_t capacity) {
  if (!e || !ms) return fail(PS_E_ARG, "null argument");
  const int R = (e->cfg.max_steps + e->cfg.replan_freq - 1) / e->cfg.replan_freq;
  if (!e->policy_events || (int)e->pev.size() < 2 * R) return fail(PS_E_STATE, "ps_policy_event_times: call ps_enable_policy_events first");
  if (capacity < R) return fail(PS_E_ARG, "destination too small");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->stream));
  for (int t = 0; t < R; ++t) HIPCHK(hipEventElapsedTime(&ms[t], e->pev[2 * t], e->pev[2 * t + 1]));
  return R;
}

extern "C" void* ps_stream(ps_engine* e) { return e ? (void*)e->stream : nullptr; }

extern "C" int ps_set_state(ps_engine* e, int32_t steps, const float* traj, const float* vel) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_set_state before ps_set_scene");
  if (steps < e->cfg.hist_steps || steps > e->stride_steps) return fail(PS_E_ARG, "steps out of range");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipMemcpy2DAsync(e->d_traj.p, sizeof(float) * 4 * e->stride_steps, traj, sizeof(float) * 4 * steps, sizeof(float) * 4 * steps,
                          e->A, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpy2DAsync(e->d_vel.p, sizeof(float) * 2 * e->stride_steps, vel, sizeof(float) * 2 * steps, sizeof(float) * 2 * steps,
                          e->A, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  e->reset = true;
  e->env_ready = -1;   // (the next replan's step_env must see the edited states)
  return PS_OK;
}

extern "C" int ps_rollout_metric(ps_engine* e, const float* gt_dev, float* out_dev) {
  if (!e || !e->reset) return fail(PS_E_STATE, "ps_rollout_metric before a rollout");
  if (!out_dev) return fail(PS_E_ARG, "null output");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq;
  hipLaunchKernelGGL(k_rollout_metric, dim3((e->A + 127) / 128), dim3(128), 0, e->stream, (const float*)e->d_traj.p, e->stride_steps,
                     c.hist_steps, R * c.replan_freq, gt_dev, e->A, out_dev, e->all_policy ? (const int*)nullptr : (const int*)e->d_is_policy.p);
  HIPCHK(hipGetLastError());
  return PS_OK;
}

extern "C" int ps_pair_metric(ps_engine* e, const float* tgt_dev, const uint8_t* pair_mask_dev, const float* prob_dev, float* out_dev) {
  if (!e || !e->reset) return fail(PS_E_STATE, "ps_pair_metric before a rollout");
  if (!tgt_dev || !pair_mask_dev || !out_dev) return fail(PS_E_ARG, "ps_pair_metric: null target / mask / output");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq;
  hipLaunchKernelGGL(k_pair_metric, dim3((e->A + 127) / 128), dim3(128), 0, e->stream, (const float*)e->d_motion.p, prob_dev, tgt_dev,
                     pair_mask_dev, R, e->A, c.motion_k, c.target_steps, c.state_dim, c.replan_freq,
                     e->all_policy ? (const int*)nullptr : (const int*)e->d_is_policy.p, out_dev);
  HIPCHK(hipGetLastError());
  return PS_OK;
}

// obtain_rollout_trajs_in_world (rollout/gpu_utils.py:230-281) on the device: [A][T][3] world-frame (x, y, heading) of the
// rolled-out steps.  center_to_world: host, row-major 3 x 3 (batch.centered_world_from_agent_tf[0]); NULL = identity.
// out_dev NULL: the result stays in the engine (ps_get "world_traj").
extern "C" int ps_world_trajs(ps_engine* e, const float* center_to_world, float* out_dev) {
  if (!e || !e->have_scene || !e->reset) return fail(PS_E_STATE, "ps_world_trajs before a rollout");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq, T = R * c.replan_freq;
  WorldTf tf{{1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f}};
  if (center_to_world) std::memcpy(tf.m, center_to_world, sizeof(tf.m));
  if (!out_dev) {
    if (e->d_world.ensure((size_t)e->A * T * 3)) return fail(PS_E_HIP, "device allocation failed");
    out_dev = e->d_world.p;
  }
  hipLaunchKernelGGL(k_world_traj, dim3((e->A * T + 255) / 256), dim3(256), 0, e->stream, (const float*)e->d_traj.p, e->stride_steps,
                     c.hist_steps, T, e->A, (const float*)e->d_init_pos.p, (const float*)e->d_init_head.p, tf, out_dev);
  HIPCHK(hipGetLastError());
  return PS_OK;
}

extern "C" int32_t ps_num_agents(ps_engine* e) { return e ? e->A : 0; }
extern "C" int32_t ps_num_map_tokens(ps_engine* e) { return e ? e->Mv : 0; }

extern "C" int64_t ps_get(ps_engine* e, const char* name, float* dst, int64_t capacity) {
  if (!e || !name || !dst) return fail(PS_E_ARG, "null argument");
  if (hipSetDevice(e->cfg.device) != hipSuccess) return fail(PS_E_HIP, "hipSetDevice");
  if (hipStreamSynchronize(e->stream) != hipSuccess) return fail(PS_E_HIP, std::string("stream error: ") + hipGetErrorString(hipGetLastError()));
  const ps_config& c = e->cfg;
  const int A = e->A, Mv = e->Mv;
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq;
  const std::string n(name);
  auto copy = [&](const float* src, int64_t count) -> int64_t {
    if (count > capacity) return fail(PS_E_ARG, "destination too small for '" + n + "'");
    if (hipMemcpyAsync(dst, src, sizeof(float) * count, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess)
      return fail(PS_E_HIP, "hipMemcpy D2H");
    return count;
  };
  if (n == "traj" || n == "vel") {
    const int w = n == "traj" ? 4 : 2;
    const int64_t steps = (int64_t)R * c.replan_freq, count = (int64_t)A * steps * w;
    if (count > capacity) return fail(PS_E_ARG, "destination too small");
    const float* src = (n == "traj" ? e->d_traj.p : e->d_vel.p) + (size_t)c.hist_steps * w;
    if (hipMemcpy2D(dst, sizeof(float) * w * steps, src, sizeof(float) * w * e->stride_steps, sizeof(float) * w * steps, A,
                    hipMemcpyDeviceToHost) != hipSuccess)
      return fail(PS_E_HIP, "hipMemcpy2D D2H");
    return count;
  }
  if (n == "world_traj") {
    if (!e->d_world.p) return fail(PS_E_STATE, "ps_get 'world_traj' before ps_world_trajs");
    return copy(e->d_world.p, (int64_t)A * R * c.replan_freq * 3);
  }
  if (n == "motion_pred") return copy(e->d_motion.p, (int64_t)R * A * c.motion_k * c.target_steps * c.state_dim);
  if (n == "reconst_pred") {
    if (c.no_reconst_pred) return fail(PS_E_ARG, "no reconst_pred: the model was built without USE_GOAL_PRED_LOSS (ps_config.no_reconst_pred)");
    return copy(e->d_reconst.p, (int64_t)A * 2);
  }
  if (n == "policy_emd") return copy(e->d_emd.p, (int64_t)A * D);
  if (n == "goal_prob" || n == "goal_point") {
    if (c.goal_pred_k <= 0) return fail(PS_E_ARG, "the engine was created without goal heads (goal_pred_k = 0)");
    return n == "goal_prob" ? copy(e->d_goal_prob.p, (int64_t)A * c.goal_pred_k) : copy(e->d_goal_point.p, (int64_t)A * 2 * c.goal_pred_k);
  }
  if (n == "scene_tokens") return copy(e->d_tok.p, (int64_t)(Mv + A) * D);
  if (n == "fused") return copy(e->d_fused.p, (int64_t)A * D);
  if (n == "obs_in") return copy(e->d_obs_in.p, (int64_t)A * c.hist_steps * c.obs_dim);
  if (n == "cur_pos") return copy(e->d_cur_pos.p, (int64_t)A * 2);
  if (n == "edge_counts") {
    if (capacity < 8) return fail(PS_E_ARG, "destination too small");
    int v[4] = {0, 0, 0, 0};
    (void)hipMemcpy(&v[0], e->e_p2p.eoff.p + e->Ap, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&v[1], e->e_s2p.eoff.p + e->Ap, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&v[2], e->e_a2p.eoff.p + A, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipMemcpy(&v[3], e->e_m2p.eoff.p + A, sizeof(int), hipMemcpyDeviceToHost);
    dst[0] = e->edge_counts[0]; dst[1] = e->edge_counts[1];
    dst[2] = (float)v[0]; dst[3] = (float)v[1]; dst[4] = (float)v[2]; dst[5] = (float)v[3];
    dst[6] = e->edge_counts[6]; dst[7] = 0.f;
    return 8;
  }
#ifdef PS_EXPERIMENTS   // (tools/gpu_stage_bisect.py: raw intermediate buffers of the encoder, as float words)
  if (n == "dbg_kv_agents") return copy(e->d_kv.p + (size_t)Mv * 256, (int64_t)e->Ap * 256);
  if (n == "dbg_kh_agents") return copy(reinterpret_cast<const float*>(e->d_kh.p + (size_t)Mv * 256), (int64_t)e->Ap * 128);
  if (n == "dbg_kv_all") return copy(e->d_kv.p, (int64_t)(Mv + e->Ap) * 256);
  if (n == "dbg_kh_all") return copy(reinterpret_cast<const float*>(e->d_kh.p), (int64_t)(Mv + e->Ap) * 128);
  if (n == "dbg_geo_a2a" || n == "dbg_geo_s2s" || n == "dbg_esrc_a2a" || n == "dbg_esrc_s2s") {
    const bool a2a = n.back() == 'a';
    EdgeSet& es = a2a ? e->e_a2a : e->e_s2s;
    int E = 0;
    (void)hipMemcpy(&E, es.eoff.p + es.nq, sizeof(int), hipMemcpyDeviceToHost);
    if (n[4] == 'g') return es.geo.p ? copy(reinterpret_cast<const float*>(es.geo.p), (int64_t)E * 8) : fail(PS_E_STATE, "no records");
    return copy(reinterpret_cast<const float*>(es.esrc.p), (int64_t)E);
  }
#endif
  return fail(PS_E_ARG, "unknown result name '" + n + "'");
}

// ps_get without the two synchronisations: the copy of a per-agent result is ENQUEUED on the engine's stream behind whatever the
// stream holds (typically right after ps_rollout) into memory the caller owns; with pinned host memory the call returns at once and
// the caller waits for the stream (or an event it records on ps_stream) before reading.  What a serving loop needs to keep an
// engine's stream non-empty: results of batch n leave while batch n + 1 is already queued behind them.
extern "C" int64_t ps_get_async(ps_engine* e, const char* name, float* dst, int64_t capacity) {
  if (!e || !name || !dst) return fail(PS_E_ARG, "null argument");
  if (!e->have_scene) return fail(PS_E_STATE, "ps_get_async before ps_set_scene");
  if (hipSetDevice(e->cfg.device) != hipSuccess) return fail(PS_E_HIP, "hipSetDevice");
  const ps_config& c = e->cfg;
  const int A = e->A;
  const int R = (c.max_steps + c.replan_freq - 1) / c.replan_freq;
  const std::string n(name);
  const float* src = nullptr;
  int64_t count = 0;
  int w2d = 0;   // traj / vel: rows of stride_steps steps, the history cut off
  if (n == "traj" || n == "vel") {
    w2d = n == "traj" ? 4 : 2;
    count = (int64_t)A * R * c.replan_freq * w2d;
    src = (n == "traj" ? e->d_traj.p : e->d_vel.p) + (size_t)c.hist_steps * w2d;
  } else if (n == "motion_pred") { src = e->d_motion.p; count = (int64_t)R * A * c.motion_k * c.target_steps * c.state_dim; }
  else if (n == "reconst_pred" && !c.no_reconst_pred) { src = e->d_reconst.p; count = (int64_t)A * 2; }
  else if (n == "policy_emd") { src = e->d_emd.p; count = (int64_t)A * D; }
  else if (n == "fused") { src = e->d_fused.p; count = (int64_t)A * D; }
  else if (n == "goal_prob" && c.goal_pred_k > 0) { src = e->d_goal_prob.p; count = (int64_t)A * c.goal_pred_k; }
  else if (n == "goal_point" && c.goal_pred_k > 0) { src = e->d_goal_point.p; count = (int64_t)A * 2 * c.goal_pred_k; }
  else return fail(PS_E_ARG, "ps_get_async: '" + n + "' is not a per-agent result of this engine");
  if (count > capacity) return fail(PS_E_ARG, "destination too small for '" + n + "'");
  if (w2d) {
    const int64_t steps = (int64_t)R * c.replan_freq;
    if (hipMemcpy2DAsync(dst, sizeof(float) * w2d * steps, src, sizeof(float) * w2d * e->stride_steps, sizeof(float) * w2d * steps, A,
                         hipMemcpyDeviceToHost, e->stream) != hipSuccess)
      return fail(PS_E_HIP, "hipMemcpy2DAsync D2H");
  } else if (hipMemcpyAsync(dst, src, sizeof(float) * (size_t)count, hipMemcpyDeviceToHost, e->stream) != hipSuccess) {
    return fail(PS_E_HIP, "hipMemcpyAsync D2H");
  }
  return count;
}

// out[0] = rollouts that captured and instantiated a graph, out[1] = rollouts after a setter that could keep the graph they had
extern "C" int ps_graph_stats(ps_engine* e, int64_t* out) {
  if (!e || !out) return fail(PS_E_ARG, "null argument");
  out[0] = e->graph_captures;
  out[1] = e->graph_reuses;
  return PS_OK;
}

extern "C" int ps_time_rollout(ps_engine* e, int32_t warmup, int32_t iters, float* ms_rollout, float* stage_ms) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_time_rollout before ps_set_scene");
  HIPCHK(hipSetDevice(e->cfg.device));
  int rc;
  for (int i = 0; i < warmup; ++i)
    if ((rc = ps_rollout(e))) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  hipEvent_t a, b, c1, c2;
  // (the per-stage split below runs the stages eagerly; ms_rollout is re-measured on ps_rollout itself)
  HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b)); HIPCHK(hipEventCreate(&c1)); HIPCHK(hipEventCreate(&c2));
  const int R = (e->cfg.max_steps + e->cfg.replan_freq - 1) / e->cfg.replan_freq;
  double tot = 0, s0 = 0, s1 = 0, s2 = 0;
  for (int i = 0; i < iters; ++i) {
    HIPCHK(hipEventRecord(a, e->stream));
    if ((rc = ps_encode_scene(e))) return rc;
    HIPCHK(hipEventRecord(c1, e->stream));
    if ((rc = ps_generate_policy(e))) return rc;
    if ((rc = ps_reset_rollout(e))) return rc;
    HIPCHK(hipEventRecord(c2, e->stream));
    for (int t = 0; t < R; ++t)
      if ((rc = ps_policy_step(e, t))) return rc;
    HIPCHK(hipEventRecord(b, e->stream));
    HIPCHK(hipEventSynchronize(b));
    float m = 0;
    HIPCHK(hipEventElapsedTime(&m, a, b)); tot += m;
    HIPCHK(hipEventElapsedTime(&m, a, c1)); s0 += m;
    HIPCHK(hipEventElapsedTime(&m, c1, c2)); s1 += m;
    HIPCHK(hipEventElapsedTime(&m, c2, b)); s2 += m;
  }
  // the product path: ps_rollout (graph replay)
  HIPCHK(hipEventRecord(a, e->stream));
  for (int i = 0; i < iters; ++i)
    if ((rc = ps_rollout(e))) return rc;
  HIPCHK(hipEventRecord(b, e->stream));
  HIPCHK(hipEventSynchronize(b));
  {
    float m = 0;
    HIPCHK(hipEventElapsedTime(&m, a, b));
    tot = m;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c1); (void)hipEventDestroy(c2);
  if (ms_rollout) *ms_rollout = (float)(tot / std::max(1, iters));
  if (stage_ms) { stage_ms[0] = (float)(s0 / std::max(1, iters)); stage_ms[1] = (float)(s1 / std::max(1, iters)); stage_ms[2] = (float)(s2 / std::max(1, iters)); }
  return PS_OK;
}

extern "C" int ps_time_policy_kernel(ps_engine* e, int32_t iters, float* ms_kernel) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_time_policy_kernel before ps_set_scene");
  e->time_chain = true;
  e->chain_ms_sum = 0;
  e->chain_launches = 0;
  int rc = 0;
  for (int i = 0; i < iters && !rc; ++i) rc = ps_rollout(e);
  e->time_chain = false;
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(e->stream));
  if (ms_kernel) *ms_kernel = (float)(e->chain_ms_sum / std::max(1, e->chain_launches));
  return PS_OK;
}

// ------------------------------------------------------------------------------------------ test hooks
extern "C" int ps_test_pointnet(ps_engine* e, int32_t which, int32_t n_poly, int32_t P, const float* x, const uint8_t* point_mask,
                                float* out) {
  if (!e) return fail(PS_E_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  const PointNetW& w = which == 0 ? e->pn_map : e->pn_obs;
  DevBuf<float> dx, dout;
  DevBuf<uint8_t> dm;
  if (upload(dx, x, (size_t)n_poly * P * w.in_dim, e->stream) || upload(dm, point_mask, (size_t)n_poly * P, e->stream) ||
      dout.ensure((size_t)n_poly * D))
    return fail(PS_E_HIP, "test upload failed");
  launch_pointnet(e, w, dx.p, dm.p, nullptr, n_poly, P, 0, dout.p);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(out, dout.p, sizeof(float) * n_poly * D, hipMemcpyDeviceToHost));
  dx.release(); dm.release(); dout.release();
  return PS_OK;
}

extern "C" int ps_test_pointnet_mt(ps_engine* e, int32_t which, int32_t n_poly, int32_t P, const float* x, const uint8_t* point_mask,
                                   float* out, int32_t mt, int32_t iters, float* ms_out) {
  if (!e) return fail(PS_E_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  const PointNetW& w = which == 0 ? e->pn_map : e->pn_obs;
  DevBuf<float> dx, dout;
  DevBuf<uint8_t> dm;
  if (upload(dx, x, (size_t)n_poly * P * w.in_dim, e->stream) || upload(dm, point_mask, (size_t)n_poly * P, e->stream) ||
      dout.ensure((size_t)n_poly * D))
    return fail(PS_E_HIP, "test upload failed");
  if (mt >= 1 && !pointnet_shape(n_poly, P, false, mt).mt)   // (a forced tiling without a build for this P used to fall back to the staged kernel silently)
    return fail(PS_E_ARG, "ps_test_pointnet_mt: no row-tile build with " + std::to_string(mt) + " tiles per wave holds " + std::to_string(P) + " points");
  e->force_mt = mt;
  launch_pointnet(e, w, dx.p, dm.p, nullptr, n_poly, P, 0, dout.p);
  HIPCHK(hipStreamSynchronize(e->stream));
  if (iters > 0 && ms_out) {
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    HIPCHK(hipEventRecord(a, e->stream));
    for (int i = 0; i < iters; ++i) launch_pointnet(e, w, dx.p, dm.p, nullptr, n_poly, P, 0, dout.p);
    HIPCHK(hipEventRecord(b, e->stream));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    *ms_out = ms / iters;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  }
  e->force_mt = 0;
#ifdef PS_RT_PROF
  {
    unsigned long long h[32];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rt_prof), sizeof(h));
    fprintf(stderr, "[rt prof] mt=%d cycles of wave 0 per launch:", mt);
    static const char* nm[10] = {"input", "gemm0", "epi0", "pool1", "to_op", "pooledgemm", "midgemm", "midepi", "pool2", "out"};
    for (int i = 0; i < 10; ++i) fprintf(stderr, " %s:%.0f", nm[i], (double)h[i] / (iters + 1));
    fprintf(stderr, "\n");
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rt_prof), z, sizeof(z));
  }
#endif
  HIPCHK(hipMemcpy(out, dout.p, sizeof(float) * n_poly * D, hipMemcpyDeviceToHost));
  dx.release(); dm.release(); dout.release();
  return PS_OK;
}

extern "C" int ps_test_fourier(ps_engine* e, int32_t n, const float* x4, float* out128) {
  if (!e) return fail(PS_E_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  DevBuf<float> dx, dout;
  if (upload(dx, x4, (size_t)n * 4, e->stream) || dout.ensure((size_t)n * 128)) return fail(PS_E_HIP, "test upload failed");
  hipLaunchKernelGGL(k_fourier_test, dim3((n * 128 + 255) / 256), dim3(256), 0, e->stream, (const float*)dx.p, n, e->div32, dout.p);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(out128, dout.p, sizeof(float) * n * 128, hipMemcpyDeviceToHost));
  dx.release(); dout.release();
  return PS_OK;
}

extern "C" int ps_test_wrap(ps_engine* e, int32_t n, const float* x, float* out) {
  if (!e) return fail(PS_E_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  DevBuf<float> dx, dout;
  if (upload(dx, x, (size_t)n, e->stream) || dout.ensure((size_t)n)) return fail(PS_E_HIP, "test upload failed");
  hipLaunchKernelGGL(k_wrap_test, dim3((n + 255) / 256), dim3(256), 0, e->stream, (const float*)dx.p, n, dout.p);
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(out, dout.p, sizeof(float) * n, hipMemcpyDeviceToHost));
  dx.release(); dout.release();
  return PS_OK;
}

extern "C" int ps_test_attn(ps_engine* e, int32_t layer_index, int32_t Ns, int32_t Nd, int32_t E, const float* x_src,
                            const float* x_dst, const float* rt, const int32_t* eoff, const int32_t* esrc, int32_t T,
                            float* out) {
  if (!e) return fail(PS_E_ARG, "null engine");
  if (layer_index < 0 || layer_index >= (int)e->all_layers.size()) return fail(PS_E_ARG, "layer index out of range");
  HIPCHK(hipSetDevice(e->cfg.device));
  DevBuf<float> dxs, dxd, drt, dkv;
  DevBuf<_Float16> drth, dkh, drtA, drtT;
  DevBuf<int> doff, dsrc, dtoff;
  DevBuf<ChainStep> dstep;
  int maxdeg = 1;
  std::vector<int> toff((size_t)Nd + 1, 0);
  for (int i = 0; i < Nd; ++i) {
    maxdeg = std::max(maxdeg, eoff[i + 1] - eoff[i]);
    toff[i + 1] = toff[i] + (eoff[i + 1] - eoff[i] + 31) / 32;
  }
  if (upload(dxs, x_src, (size_t)Ns * D, e->stream) || upload(dxd, x_dst, (size_t)Nd * D, e->stream) ||
      upload(drt, rt, (size_t)std::max(E, 1) * 128, e->stream) || upload(doff, (const int*)eoff, (size_t)Nd + 1, e->stream) ||
      upload(dsrc, (const int*)esrc, (size_t)std::max(E, 1), e->stream) || dkv.ensure((size_t)Ns * 256) ||
      drth.ensure((size_t)std::max(E, 1) * 256) || dkh.ensure((size_t)Ns * 256) ||
      upload(dtoff, toff.data(), toff.size(), e->stream) || drtA.ensure((size_t)(toff[Nd] + 1) * 8192) || drtT.ensure((size_t)(toff[Nd] + 1) * 8192))
    return fail(PS_E_HIP, "test upload failed");
  launch_kv(e, dxs.p, Ns, layer_index, 1, dkv.p, dkh.p, 0);
  hipLaunchKernelGGL(k_split_rows, dim3((std::max(E, 1) * 128 + 255) / 256), dim3(256), 0, e->stream, (const float*)drt.p, std::max(E, 1), drth.p);
  if (Nd > 0)
    hipLaunchKernelGGL(k_tile_transpose, dim3(std::max(1, std::min(toff[Nd], 2048))), dim3(256), 0, e->stream, (const int*)doff.p,
                       (const int*)dtoff.p, Nd, (const _Float16*)drth.p, drtA.p, drtT.p);
  ChainStep st;
  st.w = e->all_layers[layer_index];
  st.kv = dkv.p; st.eoff = doff.p; st.esrc = dsrc.p; st.toff = dtoff.p; st.rtT = drtT.p; st.rtA = drtA.p; st.khl = dkh.p;
  st.kr = 4;   // the hook is handed arbitrary rows: all 128 columns count
  st.geo = nullptr;
  if (upload(dstep, &st, 1, e->stream)) return fail(PS_E_HIP, "test upload failed");
  if (T == 16 && maxdeg <= ES_MAXDEG) {   // the split layer (k_node + k_edge_small + k_node)
    if (launch_split_layer(e, dxd.p, Nd, dstep.p, 4, maxdeg, nullptr, nullptr)) return PS_E_HIP;
  } else if (launch_chain(e, dxd.p, Nd, 0, 1, maxdeg, false, dstep.p, T == 16 ? 0 : T, 4)) return PS_E_HIP;
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(out, dxd.p, sizeof(float) * (size_t)Nd * D, hipMemcpyDeviceToHost));
  dxs.release(); dxd.release(); drt.release(); dkv.release(); doff.release(); dsrc.release(); dstep.release(); drth.release(); dkh.release();
  dtoff.release(); drtT.release(); drtA.release();
  return PS_OK;
}

// Debug read-back of an edge set: which = 0 a2a, 1 s2s, 2 p2p, 3 s2p, 4 a2p, 5 m2p.  Returns #edges.
extern "C" int64_t ps_test_get_edges(ps_engine* e, int32_t which, int32_t* esrc, int32_t* edst, float* rt, int64_t capacity) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "no scene");
  if (hipSetDevice(e->cfg.device) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess) return fail(PS_E_HIP, "sync");
  EdgeSet* sets[6] = {&e->e_a2a, &e->e_s2s, &e->e_p2p, &e->e_s2p, &e->e_a2p, &e->e_m2p};
  if (which < 0 || which > 5) return fail(PS_E_ARG, "bad edge set");
  EdgeSet& s = *sets[which];
  int E = 0;
  HIPCHK(hipStreamSynchronize(e->stream));
  if (hipMemcpy(&E, s.eoff.p + s.nq, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return fail(PS_E_HIP, "memcpy");
  if (E > capacity) return fail(PS_E_ARG, "capacity too small");
  if (hipMemcpy(esrc, s.esrc.p, sizeof(int) * E, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(edst, s.edst.p, sizeof(int) * E, hipMemcpyDeviceToHost) != hipSuccess)
    return fail(PS_E_HIP, "memcpy");
  if (rt) {   // the engine keeps the rows as split-fp16 operand images only: hi + lo restores them to ~2^-22
    std::vector<int> eo((size_t)s.nq + 1), to((size_t)s.nq + 1);
    if (hipMemcpy(eo.data(), s.eoff.p, sizeof(int) * eo.size(), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(to.data(), s.toff.p, sizeof(int) * to.size(), hipMemcpyDeviceToHost) != hipSuccess)
      return fail(PS_E_HIP, "memcpy");
    std::vector<_Float16> img((size_t)to[s.nq] * 8192);
    if (!img.empty() && hipMemcpy(img.data(), s.rtA.p, sizeof(_Float16) * img.size(), hipMemcpyDeviceToHost) != hipSuccess)
      return fail(PS_E_HIP, "memcpy");
    for (int d = 0; d < s.nq; ++d)
      for (int i = 0; i < eo[d + 1] - eo[d]; ++i) {
        const size_t tile = (size_t)to[d] + i / 32;
        const int sub = (i % 32) / 16, m = i % 16;
        for (int c = 0; c < 128; ++c) {
          const int cc = c < 96 ? c : c - 32;   // columns 96..127 repeat 64..95 and are not stored
          const int ks = cc / 32, kq = (cc % 32) / 8, j = cc % 8;
          const size_t hi_ = tile * 8192 + (size_t)((sub * 2 + 0) * 4 + ks) * 512 + (kq * 16 + m) * 8 + j;
          const size_t lo_ = tile * 8192 + (size_t)((sub * 2 + 1) * 4 + ks) * 512 + (kq * 16 + m) * 8 + j;
          rt[(size_t)(eo[d] + i) * 128 + c] = (float)img[hi_] + (float)img[lo_];
        }
      }
  }
  return E;
}

// ---- micro-benchmark hook: how fast can `nwg` workgroups (256 threads each) stream the SAME
// `mbytes` buffer through their CUs with plain float4 loads, `depth` x 16 loads in flight per thread?
namespace {
template <int DEPTH>
__global__ __launch_bounds__(256, 1) void k_stream_test(const float* __restrict__ buf, size_t n4, float* out) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t stride = 256;
  for (size_t i = threadIdx.x; i + (16 * DEPTH - 1) * stride < n4; i += 16 * DEPTH * stride) {
    float4 v[16 * DEPTH];
#pragma unroll
    for (int j = 0; j < 16 * DEPTH; ++j) v[j] = ldg4(buf + 4 * (i + j * stride));
#pragma unroll
    for (int j = 0; j < 16 * DEPTH; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}
}  // namespace
extern "C" int ps_test_stream(ps_engine* e, int32_t mbytes, int32_t nwg, int32_t depth, int32_t iters, float* ms_out) {
  if (!e) return fail(PS_E_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  DevBuf<float> buf, out;
  const size_t n = (size_t)mbytes * 1024 * 1024 / 4;
  if (buf.ensure(n) || out.ensure(4096)) return fail(PS_E_HIP, "alloc");
  if (dev_zero(e->stream, buf.p, n * 4)) return fail(PS_E_HIP, "device fill launch failed");
  hipEvent_t a, b;
  HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
  for (int it = -1; it < iters; ++it) {
    if (it == 0) HIPCHK(hipEventRecord(a, e->stream));
    if (depth == 1) hipLaunchKernelGGL(k_stream_test<1>, dim3(nwg), dim3(256), 0, e->stream, (const float*)buf.p, n / 4, out.p);
    else if (depth == 2) hipLaunchKernelGGL(k_stream_test<2>, dim3(nwg), dim3(256), 0, e->stream, (const float*)buf.p, n / 4, out.p);
    else hipLaunchKernelGGL(k_stream_test<3>, dim3(nwg), dim3(256), 0, e->stream, (const float*)buf.p, n / 4, out.p);
  }
  HIPCHK(hipEventRecord(b, e->stream));
  HIPCHK(hipEventSynchronize(b));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, a, b));
  *ms_out = ms / iters;
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  buf.release(); out.release();
  return PS_OK;
}


#ifdef PS_EXPERIMENTS
// ---- the "poison" load: every workgroup fills the 160 KB of LDS of its CU and the vector / accumulator registers of its
// waves with one bit pattern and leaves.  A kernel of ANOTHER stream that reads LDS or registers it never wrote then finds this pattern instead
// of its own leftovers: a run-to-run difference that only shows under load becomes deterministic (tools/gpu_stage_stress.py PS_POISON=1).
namespace {
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_poison(unsigned pat, unsigned* sink) {
  extern __shared__ unsigned poison_lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) poison_lds[i] = pat;
  __syncthreads();
  if (poison_lds[(threadIdx.x * 97) % (160 * 1024 / 4)] != pat) sink[0] = 1;   // (keeps the stores)
  asm volatile("v_mov_b32 v8, %0\n\t"
               "v_mov_b32 v9, %0\n\t"
               "v_mov_b32 v10, %0\n\t"
               "v_mov_b32 v11, %0\n\t"
               "v_mov_b32 v12, %0\n\t"
               "v_mov_b32 v13, %0\n\t"
               "v_mov_b32 v14, %0\n\t"
               "v_mov_b32 v15, %0\n\t"
               "v_mov_b32 v16, %0\n\t"
               "v_mov_b32 v17, %0\n\t"
               "v_mov_b32 v18, %0\n\t"
               "v_mov_b32 v19, %0\n\t"
               "v_mov_b32 v20, %0\n\t"
               "v_mov_b32 v21, %0\n\t"
               "v_mov_b32 v22, %0\n\t"
               "v_mov_b32 v23, %0\n\t"
               "v_mov_b32 v24, %0\n\t"
               "v_mov_b32 v25, %0\n\t"
               "v_mov_b32 v26, %0\n\t"
               "v_mov_b32 v27, %0\n\t"
               "v_mov_b32 v28, %0\n\t"
               "v_mov_b32 v29, %0\n\t"
               "v_mov_b32 v30, %0\n\t"
               "v_mov_b32 v31, %0\n\t"
               "v_mov_b32 v32, %0\n\t"
               "v_mov_b32 v33, %0\n\t"
               "v_mov_b32 v34, %0\n\t"
               "v_mov_b32 v35, %0\n\t"
               "v_mov_b32 v36, %0\n\t"
               "v_mov_b32 v37, %0\n\t"
               "v_mov_b32 v38, %0\n\t"
               "v_mov_b32 v39, %0\n\t"
               "v_mov_b32 v40, %0\n\t"
               "v_mov_b32 v41, %0\n\t"
               "v_mov_b32 v42, %0\n\t"
               "v_mov_b32 v43, %0\n\t"
               "v_mov_b32 v44, %0\n\t"
               "v_mov_b32 v45, %0\n\t"
               "v_mov_b32 v46, %0\n\t"
               "v_mov_b32 v47, %0\n\t"
               "v_mov_b32 v48, %0\n\t"
               "v_mov_b32 v49, %0\n\t"
               "v_mov_b32 v50, %0\n\t"
               "v_mov_b32 v51, %0\n\t"
               "v_mov_b32 v52, %0\n\t"
               "v_mov_b32 v53, %0\n\t"
               "v_mov_b32 v54, %0\n\t"
               "v_mov_b32 v55, %0\n\t"
               "v_mov_b32 v56, %0\n\t"
               "v_mov_b32 v57, %0\n\t"
               "v_mov_b32 v58, %0\n\t"
               "v_mov_b32 v59, %0\n\t"
               "v_mov_b32 v60, %0\n\t"
               "v_mov_b32 v61, %0\n\t"
               "v_mov_b32 v62, %0\n\t"
               "v_mov_b32 v63, %0\n\t"
               "v_mov_b32 v64, %0\n\t"
               "v_mov_b32 v65, %0\n\t"
               "v_mov_b32 v66, %0\n\t"
               "v_mov_b32 v67, %0\n\t"
               "v_mov_b32 v68, %0\n\t"
               "v_mov_b32 v69, %0\n\t"
               "v_mov_b32 v70, %0\n\t"
               "v_mov_b32 v71, %0\n\t"
               "v_mov_b32 v72, %0\n\t"
               "v_mov_b32 v73, %0\n\t"
               "v_mov_b32 v74, %0\n\t"
               "v_mov_b32 v75, %0\n\t"
               "v_mov_b32 v76, %0\n\t"
               "v_mov_b32 v77, %0\n\t"
               "v_mov_b32 v78, %0\n\t"
               "v_mov_b32 v79, %0\n\t"
               "v_mov_b32 v80, %0\n\t"
               "v_mov_b32 v81, %0\n\t"
               "v_mov_b32 v82, %0\n\t"
               "v_mov_b32 v83, %0\n\t"
               "v_mov_b32 v84, %0\n\t"
               "v_mov_b32 v85, %0\n\t"
               "v_mov_b32 v86, %0\n\t"
               "v_mov_b32 v87, %0\n\t"
               "v_mov_b32 v88, %0\n\t"
               "v_mov_b32 v89, %0\n\t"
               "v_mov_b32 v90, %0\n\t"
               "v_mov_b32 v91, %0\n\t"
               "v_mov_b32 v92, %0\n\t"
               "v_mov_b32 v93, %0\n\t"
               "v_mov_b32 v94, %0\n\t"
               "v_mov_b32 v95, %0\n\t"
               "v_mov_b32 v96, %0\n\t"
               "v_mov_b32 v97, %0\n\t"
               "v_mov_b32 v98, %0\n\t"
               "v_mov_b32 v99, %0\n\t"
               "v_mov_b32 v100, %0\n\t"
               "v_mov_b32 v101, %0\n\t"
               "v_mov_b32 v102, %0\n\t"
               "v_mov_b32 v103, %0\n\t"
               "v_mov_b32 v104, %0\n\t"
               "v_mov_b32 v105, %0\n\t"
               "v_mov_b32 v106, %0\n\t"
               "v_mov_b32 v107, %0\n\t"
               "v_mov_b32 v108, %0\n\t"
               "v_mov_b32 v109, %0\n\t"
               "v_mov_b32 v110, %0\n\t"
               "v_mov_b32 v111, %0\n\t"
               "v_mov_b32 v112, %0\n\t"
               "v_mov_b32 v113, %0\n\t"
               "v_mov_b32 v114, %0\n\t"
               "v_mov_b32 v115, %0\n\t"
               "v_mov_b32 v116, %0\n\t"
               "v_mov_b32 v117, %0\n\t"
               "v_mov_b32 v118, %0\n\t"
               "v_mov_b32 v119, %0\n\t"
               "v_mov_b32 v120, %0\n\t"
               "v_mov_b32 v121, %0\n\t"
               "v_mov_b32 v122, %0\n\t"
               "v_mov_b32 v123, %0\n\t"
               "v_mov_b32 v124, %0\n\t"
               "v_mov_b32 v125, %0\n\t"
               "v_mov_b32 v126, %0\n\t"
               "v_mov_b32 v127, %0\n\t"
               "v_mov_b32 v128, %0\n\t"
               "v_mov_b32 v129, %0\n\t"
               "v_mov_b32 v130, %0\n\t"
               "v_mov_b32 v131, %0\n\t"
               "v_mov_b32 v132, %0\n\t"
               "v_mov_b32 v133, %0\n\t"
               "v_mov_b32 v134, %0\n\t"
               "v_mov_b32 v135, %0\n\t"
               "v_mov_b32 v136, %0\n\t"
               "v_mov_b32 v137, %0\n\t"
               "v_mov_b32 v138, %0\n\t"
               "v_mov_b32 v139, %0\n\t"
               "v_mov_b32 v140, %0\n\t"
               "v_mov_b32 v141, %0\n\t"
               "v_mov_b32 v142, %0\n\t"
               "v_mov_b32 v143, %0\n\t"
               "v_mov_b32 v144, %0\n\t"
               "v_mov_b32 v145, %0\n\t"
               "v_mov_b32 v146, %0\n\t"
               "v_mov_b32 v147, %0\n\t"
               "v_mov_b32 v148, %0\n\t"
               "v_mov_b32 v149, %0\n\t"
               "v_mov_b32 v150, %0\n\t"
               "v_mov_b32 v151, %0\n\t"
               "v_mov_b32 v152, %0\n\t"
               "v_mov_b32 v153, %0\n\t"
               "v_mov_b32 v154, %0\n\t"
               "v_mov_b32 v155, %0\n\t"
               "v_mov_b32 v156, %0\n\t"
               "v_mov_b32 v157, %0\n\t"
               "v_mov_b32 v158, %0\n\t"
               "v_mov_b32 v159, %0\n\t"
               "v_mov_b32 v160, %0\n\t"
               "v_mov_b32 v161, %0\n\t"
               "v_mov_b32 v162, %0\n\t"
               "v_mov_b32 v163, %0\n\t"
               "v_mov_b32 v164, %0\n\t"
               "v_mov_b32 v165, %0\n\t"
               "v_mov_b32 v166, %0\n\t"
               "v_mov_b32 v167, %0\n\t"
               "v_mov_b32 v168, %0\n\t"
               "v_mov_b32 v169, %0\n\t"
               "v_mov_b32 v170, %0\n\t"
               "v_mov_b32 v171, %0\n\t"
               "v_mov_b32 v172, %0\n\t"
               "v_mov_b32 v173, %0\n\t"
               "v_mov_b32 v174, %0\n\t"
               "v_mov_b32 v175, %0\n\t"
               "v_mov_b32 v176, %0\n\t"
               "v_mov_b32 v177, %0\n\t"
               "v_mov_b32 v178, %0\n\t"
               "v_mov_b32 v179, %0\n\t"
               "v_mov_b32 v180, %0\n\t"
               "v_mov_b32 v181, %0\n\t"
               "v_mov_b32 v182, %0\n\t"
               "v_mov_b32 v183, %0\n\t"
               "v_mov_b32 v184, %0\n\t"
               "v_mov_b32 v185, %0\n\t"
               "v_mov_b32 v186, %0\n\t"
               "v_mov_b32 v187, %0\n\t"
               "v_mov_b32 v188, %0\n\t"
               "v_mov_b32 v189, %0\n\t"
               "v_mov_b32 v190, %0\n\t"
               "v_mov_b32 v191, %0\n\t"
               "v_mov_b32 v192, %0\n\t"
               "v_mov_b32 v193, %0\n\t"
               "v_mov_b32 v194, %0\n\t"
               "v_mov_b32 v195, %0\n\t"
               "v_mov_b32 v196, %0\n\t"
               "v_mov_b32 v197, %0\n\t"
               "v_mov_b32 v198, %0\n\t"
               "v_mov_b32 v199, %0\n\t"
               "v_mov_b32 v200, %0\n\t"
               "v_mov_b32 v201, %0\n\t"
               "v_mov_b32 v202, %0\n\t"
               "v_mov_b32 v203, %0\n\t"
               "v_mov_b32 v204, %0\n\t"
               "v_mov_b32 v205, %0\n\t"
               "v_mov_b32 v206, %0\n\t"
               "v_mov_b32 v207, %0\n\t"
               "v_mov_b32 v208, %0\n\t"
               "v_mov_b32 v209, %0\n\t"
               "v_mov_b32 v210, %0\n\t"
               "v_mov_b32 v211, %0\n\t"
               "v_mov_b32 v212, %0\n\t"
               "v_mov_b32 v213, %0\n\t"
               "v_mov_b32 v214, %0\n\t"
               "v_mov_b32 v215, %0\n\t"
               "v_mov_b32 v216, %0\n\t"
               "v_mov_b32 v217, %0\n\t"
               "v_mov_b32 v218, %0\n\t"
               "v_mov_b32 v219, %0\n\t"
               "v_mov_b32 v220, %0\n\t"
               "v_mov_b32 v221, %0\n\t"
               "v_mov_b32 v222, %0\n\t"
               "v_mov_b32 v223, %0\n\t"
               "v_mov_b32 v224, %0\n\t"
               "v_mov_b32 v225, %0\n\t"
               "v_mov_b32 v226, %0\n\t"
               "v_mov_b32 v227, %0\n\t"
               "v_mov_b32 v228, %0\n\t"
               "v_mov_b32 v229, %0\n\t"
               "v_mov_b32 v230, %0\n\t"
               "v_mov_b32 v231, %0\n\t"
               "v_mov_b32 v232, %0\n\t"
               "v_mov_b32 v233, %0\n\t"
               "v_mov_b32 v234, %0\n\t"
               "v_mov_b32 v235, %0\n\t"
               "v_mov_b32 v236, %0\n\t"
               "v_mov_b32 v237, %0\n\t"
               "v_mov_b32 v238, %0\n\t"
               "v_mov_b32 v239, %0\n\t"
               "v_mov_b32 v240, %0\n\t"
               "v_mov_b32 v241, %0\n\t"
               "v_mov_b32 v242, %0\n\t"
               "v_mov_b32 v243, %0\n\t"
               "v_mov_b32 v244, %0\n\t"
               "v_mov_b32 v245, %0\n\t"
               "v_mov_b32 v246, %0\n\t"
               "v_mov_b32 v247, %0\n\t"
               "v_mov_b32 v248, %0\n\t"
               "v_mov_b32 v249, %0\n\t"
               "v_mov_b32 v250, %0\n\t"
               "v_mov_b32 v251, %0\n\t"
               "v_mov_b32 v252, %0\n\t"
               "v_mov_b32 v253, %0\n\t"
               "v_mov_b32 v254, %0\n\t"
               "v_mov_b32 v255, %0\n\t"
               "v_accvgpr_write_b32 a0, %0\n\t"
               "v_accvgpr_write_b32 a1, %0\n\t"
               "v_accvgpr_write_b32 a2, %0\n\t"
               "v_accvgpr_write_b32 a3, %0\n\t"
               "v_accvgpr_write_b32 a4, %0\n\t"
               "v_accvgpr_write_b32 a5, %0\n\t"
               "v_accvgpr_write_b32 a6, %0\n\t"
               "v_accvgpr_write_b32 a7, %0\n\t"
               "v_accvgpr_write_b32 a8, %0\n\t"
               "v_accvgpr_write_b32 a9, %0\n\t"
               "v_accvgpr_write_b32 a10, %0\n\t"
               "v_accvgpr_write_b32 a11, %0\n\t"
               "v_accvgpr_write_b32 a12, %0\n\t"
               "v_accvgpr_write_b32 a13, %0\n\t"
               "v_accvgpr_write_b32 a14, %0\n\t"
               "v_accvgpr_write_b32 a15, %0\n\t"
               "v_accvgpr_write_b32 a16, %0\n\t"
               "v_accvgpr_write_b32 a17, %0\n\t"
               "v_accvgpr_write_b32 a18, %0\n\t"
               "v_accvgpr_write_b32 a19, %0\n\t"
               "v_accvgpr_write_b32 a20, %0\n\t"
               "v_accvgpr_write_b32 a21, %0\n\t"
               "v_accvgpr_write_b32 a22, %0\n\t"
               "v_accvgpr_write_b32 a23, %0\n\t"
               "v_accvgpr_write_b32 a24, %0\n\t"
               "v_accvgpr_write_b32 a25, %0\n\t"
               "v_accvgpr_write_b32 a26, %0\n\t"
               "v_accvgpr_write_b32 a27, %0\n\t"
               "v_accvgpr_write_b32 a28, %0\n\t"
               "v_accvgpr_write_b32 a29, %0\n\t"
               "v_accvgpr_write_b32 a30, %0\n\t"
               "v_accvgpr_write_b32 a31, %0\n\t"
               "v_accvgpr_write_b32 a32, %0\n\t"
               "v_accvgpr_write_b32 a33, %0\n\t"
               "v_accvgpr_write_b32 a34, %0\n\t"
               "v_accvgpr_write_b32 a35, %0\n\t"
               "v_accvgpr_write_b32 a36, %0\n\t"
               "v_accvgpr_write_b32 a37, %0\n\t"
               "v_accvgpr_write_b32 a38, %0\n\t"
               "v_accvgpr_write_b32 a39, %0\n\t"
               "v_accvgpr_write_b32 a40, %0\n\t"
               "v_accvgpr_write_b32 a41, %0\n\t"
               "v_accvgpr_write_b32 a42, %0\n\t"
               "v_accvgpr_write_b32 a43, %0\n\t"
               "v_accvgpr_write_b32 a44, %0\n\t"
               "v_accvgpr_write_b32 a45, %0\n\t"
               "v_accvgpr_write_b32 a46, %0\n\t"
               "v_accvgpr_write_b32 a47, %0\n\t"
               "v_accvgpr_write_b32 a48, %0\n\t"
               "v_accvgpr_write_b32 a49, %0\n\t"
               "v_accvgpr_write_b32 a50, %0\n\t"
               "v_accvgpr_write_b32 a51, %0\n\t"
               "v_accvgpr_write_b32 a52, %0\n\t"
               "v_accvgpr_write_b32 a53, %0\n\t"
               "v_accvgpr_write_b32 a54, %0\n\t"
               "v_accvgpr_write_b32 a55, %0\n\t"
               "v_accvgpr_write_b32 a56, %0\n\t"
               "v_accvgpr_write_b32 a57, %0\n\t"
               "v_accvgpr_write_b32 a58, %0\n\t"
               "v_accvgpr_write_b32 a59, %0\n\t"
               "v_accvgpr_write_b32 a60, %0\n\t"
               "v_accvgpr_write_b32 a61, %0\n\t"
               "v_accvgpr_write_b32 a62, %0\n\t"
               "v_accvgpr_write_b32 a63, %0\n\t"
               "v_accvgpr_write_b32 a64, %0\n\t"
               "v_accvgpr_write_b32 a65, %0\n\t"
               "v_accvgpr_write_b32 a66, %0\n\t"
               "v_accvgpr_write_b32 a67, %0\n\t"
               "v_accvgpr_write_b32 a68, %0\n\t"
               "v_accvgpr_write_b32 a69, %0\n\t"
               "v_accvgpr_write_b32 a70, %0\n\t"
               "v_accvgpr_write_b32 a71, %0\n\t"
               "v_accvgpr_write_b32 a72, %0\n\t"
               "v_accvgpr_write_b32 a73, %0\n\t"
               "v_accvgpr_write_b32 a74, %0\n\t"
               "v_accvgpr_write_b32 a75, %0\n\t"
               "v_accvgpr_write_b32 a76, %0\n\t"
               "v_accvgpr_write_b32 a77, %0\n\t"
               "v_accvgpr_write_b32 a78, %0\n\t"
               "v_accvgpr_write_b32 a79, %0\n\t"
               "v_accvgpr_write_b32 a80, %0\n\t"
               "v_accvgpr_write_b32 a81, %0\n\t"
               "v_accvgpr_write_b32 a82, %0\n\t"
               "v_accvgpr_write_b32 a83, %0\n\t"
               "v_accvgpr_write_b32 a84, %0\n\t"
               "v_accvgpr_write_b32 a85, %0\n\t"
               "v_accvgpr_write_b32 a86, %0\n\t"
               "v_accvgpr_write_b32 a87, %0\n\t"
               "v_accvgpr_write_b32 a88, %0\n\t"
               "v_accvgpr_write_b32 a89, %0\n\t"
               "v_accvgpr_write_b32 a90, %0\n\t"
               "v_accvgpr_write_b32 a91, %0\n\t"
               "v_accvgpr_write_b32 a92, %0\n\t"
               "v_accvgpr_write_b32 a93, %0\n\t"
               "v_accvgpr_write_b32 a94, %0\n\t"
               "v_accvgpr_write_b32 a95, %0\n\t"
               "v_accvgpr_write_b32 a96, %0\n\t"
               "v_accvgpr_write_b32 a97, %0\n\t"
               "v_accvgpr_write_b32 a98, %0\n\t"
               "v_accvgpr_write_b32 a99, %0\n\t"
               "v_accvgpr_write_b32 a100, %0\n\t"
               "v_accvgpr_write_b32 a101, %0\n\t"
               "v_accvgpr_write_b32 a102, %0\n\t"
               "v_accvgpr_write_b32 a103, %0\n\t"
               "v_accvgpr_write_b32 a104, %0\n\t"
               "v_accvgpr_write_b32 a105, %0\n\t"
               "v_accvgpr_write_b32 a106, %0\n\t"
               "v_accvgpr_write_b32 a107, %0\n\t"
               "v_accvgpr_write_b32 a108, %0\n\t"
               "v_accvgpr_write_b32 a109, %0\n\t"
               "v_accvgpr_write_b32 a110, %0\n\t"
               "v_accvgpr_write_b32 a111, %0\n\t"
               "v_accvgpr_write_b32 a112, %0\n\t"
               "v_accvgpr_write_b32 a113, %0\n\t"
               "v_accvgpr_write_b32 a114, %0\n\t"
               "v_accvgpr_write_b32 a115, %0\n\t"
               "v_accvgpr_write_b32 a116, %0\n\t"
               "v_accvgpr_write_b32 a117, %0\n\t"
               "v_accvgpr_write_b32 a118, %0\n\t"
               "v_accvgpr_write_b32 a119, %0\n\t"
               "v_accvgpr_write_b32 a120, %0\n\t"
               "v_accvgpr_write_b32 a121, %0\n\t"
               "v_accvgpr_write_b32 a122, %0\n\t"
               "v_accvgpr_write_b32 a123, %0\n\t"
               "v_accvgpr_write_b32 a124, %0\n\t"
               "v_accvgpr_write_b32 a125, %0\n\t"
               "v_accvgpr_write_b32 a126, %0\n\t"
               "v_accvgpr_write_b32 a127, %0\n\t"
               "v_accvgpr_write_b32 a128, %0\n\t"
               "v_accvgpr_write_b32 a129, %0\n\t"
               "v_accvgpr_write_b32 a130, %0\n\t"
               "v_accvgpr_write_b32 a131, %0\n\t"
               "v_accvgpr_write_b32 a132, %0\n\t"
               "v_accvgpr_write_b32 a133, %0\n\t"
               "v_accvgpr_write_b32 a134, %0\n\t"
               "v_accvgpr_write_b32 a135, %0\n\t"
               "v_accvgpr_write_b32 a136, %0\n\t"
               "v_accvgpr_write_b32 a137, %0\n\t"
               "v_accvgpr_write_b32 a138, %0\n\t"
               "v_accvgpr_write_b32 a139, %0\n\t"
               "v_accvgpr_write_b32 a140, %0\n\t"
               "v_accvgpr_write_b32 a141, %0\n\t"
               "v_accvgpr_write_b32 a142, %0\n\t"
               "v_accvgpr_write_b32 a143, %0\n\t"
               "v_accvgpr_write_b32 a144, %0\n\t"
               "v_accvgpr_write_b32 a145, %0\n\t"
               "v_accvgpr_write_b32 a146, %0\n\t"
               "v_accvgpr_write_b32 a147, %0\n\t"
               "v_accvgpr_write_b32 a148, %0\n\t"
               "v_accvgpr_write_b32 a149, %0\n\t"
               "v_accvgpr_write_b32 a150, %0\n\t"
               "v_accvgpr_write_b32 a151, %0\n\t"
               "v_accvgpr_write_b32 a152, %0\n\t"
               "v_accvgpr_write_b32 a153, %0\n\t"
               "v_accvgpr_write_b32 a154, %0\n\t"
               "v_accvgpr_write_b32 a155, %0\n\t"
               "v_accvgpr_write_b32 a156, %0\n\t"
               "v_accvgpr_write_b32 a157, %0\n\t"
               "v_accvgpr_write_b32 a158, %0\n\t"
               "v_accvgpr_write_b32 a159, %0\n\t"
               "v_accvgpr_write_b32 a160, %0\n\t"
               "v_accvgpr_write_b32 a161, %0\n\t"
               "v_accvgpr_write_b32 a162, %0\n\t"
               "v_accvgpr_write_b32 a163, %0\n\t"
               "v_accvgpr_write_b32 a164, %0\n\t"
               "v_accvgpr_write_b32 a165, %0\n\t"
               "v_accvgpr_write_b32 a166, %0\n\t"
               "v_accvgpr_write_b32 a167, %0\n\t"
               "v_accvgpr_write_b32 a168, %0\n\t"
               "v_accvgpr_write_b32 a169, %0\n\t"
               "v_accvgpr_write_b32 a170, %0\n\t"
               "v_accvgpr_write_b32 a171, %0\n\t"
               "v_accvgpr_write_b32 a172, %0\n\t"
               "v_accvgpr_write_b32 a173, %0\n\t"
               "v_accvgpr_write_b32 a174, %0\n\t"
               "v_accvgpr_write_b32 a175, %0\n\t"
               "v_accvgpr_write_b32 a176, %0\n\t"
               "v_accvgpr_write_b32 a177, %0\n\t"
               "v_accvgpr_write_b32 a178, %0\n\t"
               "v_accvgpr_write_b32 a179, %0\n\t"
               "v_accvgpr_write_b32 a180, %0\n\t"
               "v_accvgpr_write_b32 a181, %0\n\t"
               "v_accvgpr_write_b32 a182, %0\n\t"
               "v_accvgpr_write_b32 a183, %0\n\t"
               "v_accvgpr_write_b32 a184, %0\n\t"
               "v_accvgpr_write_b32 a185, %0\n\t"
               "v_accvgpr_write_b32 a186, %0\n\t"
               "v_accvgpr_write_b32 a187, %0\n\t"
               "v_accvgpr_write_b32 a188, %0\n\t"
               "v_accvgpr_write_b32 a189, %0\n\t"
               "v_accvgpr_write_b32 a190, %0\n\t"
               "v_accvgpr_write_b32 a191, %0\n\t"
               "v_accvgpr_write_b32 a192, %0\n\t"
               "v_accvgpr_write_b32 a193, %0\n\t"
               "v_accvgpr_write_b32 a194, %0\n\t"
               "v_accvgpr_write_b32 a195, %0\n\t"
               "v_accvgpr_write_b32 a196, %0\n\t"
               "v_accvgpr_write_b32 a197, %0\n\t"
               "v_accvgpr_write_b32 a198, %0\n\t"
               "v_accvgpr_write_b32 a199, %0\n\t"
               "v_accvgpr_write_b32 a200, %0\n\t"
               "v_accvgpr_write_b32 a201, %0\n\t"
               "v_accvgpr_write_b32 a202, %0\n\t"
               "v_accvgpr_write_b32 a203, %0\n\t"
               "v_accvgpr_write_b32 a204, %0\n\t"
               "v_accvgpr_write_b32 a205, %0\n\t"
               "v_accvgpr_write_b32 a206, %0\n\t"
               "v_accvgpr_write_b32 a207, %0\n\t"
               "v_accvgpr_write_b32 a208, %0\n\t"
               "v_accvgpr_write_b32 a209, %0\n\t"
               "v_accvgpr_write_b32 a210, %0\n\t"
               "v_accvgpr_write_b32 a211, %0\n\t"
               "v_accvgpr_write_b32 a212, %0\n\t"
               "v_accvgpr_write_b32 a213, %0\n\t"
               "v_accvgpr_write_b32 a214, %0\n\t"
               "v_accvgpr_write_b32 a215, %0\n\t"
               "v_accvgpr_write_b32 a216, %0\n\t"
               "v_accvgpr_write_b32 a217, %0\n\t"
               "v_accvgpr_write_b32 a218, %0\n\t"
               "v_accvgpr_write_b32 a219, %0\n\t"
               "v_accvgpr_write_b32 a220, %0\n\t"
               "v_accvgpr_write_b32 a221, %0\n\t"
               "v_accvgpr_write_b32 a222, %0\n\t"
               "v_accvgpr_write_b32 a223, %0\n\t"
               "v_accvgpr_write_b32 a224, %0\n\t"
               "v_accvgpr_write_b32 a225, %0\n\t"
               "v_accvgpr_write_b32 a226, %0\n\t"
               "v_accvgpr_write_b32 a227, %0\n\t"
               "v_accvgpr_write_b32 a228, %0\n\t"
               "v_accvgpr_write_b32 a229, %0\n\t"
               "v_accvgpr_write_b32 a230, %0\n\t"
               "v_accvgpr_write_b32 a231, %0\n\t"
               "v_accvgpr_write_b32 a232, %0\n\t"
               "v_accvgpr_write_b32 a233, %0\n\t"
               "v_accvgpr_write_b32 a234, %0\n\t"
               "v_accvgpr_write_b32 a235, %0\n\t"
               "v_accvgpr_write_b32 a236, %0\n\t"
               "v_accvgpr_write_b32 a237, %0\n\t"
               "v_accvgpr_write_b32 a238, %0\n\t"
               "v_accvgpr_write_b32 a239, %0\n\t"
               "v_accvgpr_write_b32 a240, %0\n\t"
               "v_accvgpr_write_b32 a241, %0\n\t"
               "v_accvgpr_write_b32 a242, %0\n\t"
               "v_accvgpr_write_b32 a243, %0\n\t"
               "v_accvgpr_write_b32 a244, %0\n\t"
               "v_accvgpr_write_b32 a245, %0\n\t"
               "v_accvgpr_write_b32 a246, %0\n\t"
               "v_accvgpr_write_b32 a247, %0\n\t"
               "v_accvgpr_write_b32 a248, %0\n\t"
               "v_accvgpr_write_b32 a249, %0\n\t"
               "v_accvgpr_write_b32 a250, %0\n\t"
               "v_accvgpr_write_b32 a251, %0\n\t"
               "v_accvgpr_write_b32 a252, %0\n\t"
               "v_accvgpr_write_b32 a253, %0\n\t"
               "v_accvgpr_write_b32 a254, %0\n\t"
               "v_accvgpr_write_b32 a255, %0\n\t"
               ""
               :: "s"(pat) : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
}
}  // namespace
extern "C" int ps_test_poison(ps_engine* e, int32_t n_wg, uint32_t pattern, int32_t launches) {
  if (!e) return fail(PS_E_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  static unsigned* sink = nullptr;
  if (!sink) HIPCHK(hipMalloc(&sink, 256));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_poison), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(k_poison, dim3(n_wg), dim3(256), 160 * 1024, e->stream, pattern, sink);
  HIPCHK(hipGetLastError());
  return PS_OK;
}
#endif

#ifdef PS_EXPERIMENTS
// ---- round 6 hunt: geo_record itself as a probe.  Every thread makes the record of ONE fixed (source, destination) pair again and again and counts the
// iterations whose bits differ from its first one, by lane quarter: [0..3] statistics (rstd | nmr), [4..7] the three inputs.  Launched beside real
// rollouts of other engines (tools/gpu_geo_probe.py): k_edge_geo's records were found to differ under load, lanes 48-63 only, packed form only.
namespace {
__global__ __launch_bounds__(GEO_THREADS) void k_geo_probe(const float* __restrict__ src_pos, const float* __restrict__ src_ori, int ns,
                                                           const float* __restrict__ div32, float eps, int iters, unsigned long long* cnt) {
  float dv[16], rdv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    dv[k] = div32[2 * k];
    rdv[k] = 1.0f / dv[k];
  }
  const int t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
  const int s = (t * 7 + 3) % ns, d = (t * 13 + 5) % ns;
  float px = src_pos[2 * d], py = src_pos[2 * d + 1], od = src_ori[d];
  const EdgeGeo first = geo_record(s, px, py, od, cosf(od), sinf(od), src_pos, src_ori, dv, rdv, eps, 0);
  unsigned bad_stat = 0, bad_in = 0;
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(px), "+v"(py), "+v"(od));
    const EdgeGeo g = geo_record(s, px, py, od, cosf(od), sinf(od), src_pos, src_ori, dv, rdv, eps, 0);
    bad_stat += (__float_as_uint(g.rstd) != __float_as_uint(first.rstd) || __float_as_uint(g.nmr) != __float_as_uint(first.nmr)) ? 1u : 0u;
    bad_in += (__float_as_uint(g.a0) != __float_as_uint(first.a0) || __float_as_uint(g.a1) != __float_as_uint(first.a1) || __float_as_uint(g.a2) != __float_as_uint(first.a2)) ? 1u : 0u;
  }
  if (bad_stat) atomicAdd(cnt + (lane >> 4), (unsigned long long)bad_stat);
  if (bad_in) atomicAdd(cnt + 4 + (lane >> 4), (unsigned long long)bad_in);
}
}  // namespace
extern "C" int ps_test_geo_probe(ps_engine* e, int32_t n_wg, int32_t iters, int32_t launches, uint64_t* counts8) {
  if (!e || !e->have_scene) return fail(PS_E_STATE, "ps_test_geo_probe needs a scene (its token poses are the inputs)");
  HIPCHK(hipSetDevice(e->cfg.device));
  DevBuf<unsigned long long> cnt;
  if (cnt.ensure(8)) return fail(PS_E_HIP, "alloc");
  if (dev_zero(e->stream, cnt.p, 8 * sizeof(unsigned long long))) return fail(PS_E_HIP, "device fill launch failed");
  for (int i = 0; i < launches; ++i)
    hipLaunchKernelGGL(k_geo_probe, dim3(n_wg), dim3(GEO_THREADS), GEO_LDS_BYTES, e->stream, (const float*)e->d_tok_pos.p, (const float*)e->d_tok_ori.p, e->Mv + e->A,
                       e->div32, e->cfg.ln_eps, iters, cnt.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(counts8, cnt.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return PS_OK;
}
#endif
// ------------------------------------------------------------------------------------------
// Stateless policy.forward: Policy_RelPE_Temporal.forward(policy_emd, batch_obs, batch_map, batch_pos,
// pair_names, latent_state) (policy/base.py:19 -> temporal_ar.py:75-92 -> act_decoder.py:239-283, :78-140)
// on caller-supplied tokens.  Token arrays are the flattened VALID tokens in the reference's order
// (scene-major): agents [Na,128] + pos/ori/scene, map [Nm,128] + pos/ori/scene, policies [A,128] + pose/type/scene.
// Returns motion_pred [A, K, steps, state_dim] (cum-xy, wrapped cum-theta, vx, vy) and the fused feature.
extern "C" int ps_policy_forward(ps_engine* e, int32_t n_scenes, int32_t Na, const float* a_tok, const float* a_pos,
                                 const float* a_ori, const int32_t* a_scene, int32_t Nm, const float* m_tok,
                                 const float* m_pos, const float* m_ori, const int32_t* m_scene, int32_t A,
                                 const float* p_emd, const float* p_pos, const float* p_ori, const int32_t* p_type,
                                 const int32_t* p_scene, float* motion_pred, float* fused_out) {
  if (!e) return fail(PS_E_ARG, "null engine");
  if (A < 1 || Na < 0 || Nm < 0 || n_scenes < 1) return fail(PS_E_ARG, "bad sizes");
  HIPCHK(hipSetDevice(e->cfg.device));
  const ps_config& c = e->cfg;
  hipStream_t st = e->stream;
  // token geometry in the engine's [map ; agents] order
  std::vector<float> pos((size_t)(Nm + Na) * 2), ori(Nm + Na);
  std::vector<int> r_map(n_scenes + 1, 0), r_agent(n_scenes + 1, 0), pscene(A);
  for (int i = 0; i < Nm; ++i) {
    pos[2 * i] = m_pos[2 * i]; pos[2 * i + 1] = m_pos[2 * i + 1]; ori[i] = m_ori[i];
    if (m_scene[i] < 0 || m_scene[i] >= n_scenes || (i && m_scene[i] < m_scene[i - 1])) return fail(PS_E_ARG, "map tokens must be scene-major");
    r_map[m_scene[i] + 1]++;
  }
  for (int i = 0; i < Na; ++i) {
    pos[2 * (Nm + i)] = a_pos[2 * i]; pos[2 * (Nm + i) + 1] = a_pos[2 * i + 1]; ori[Nm + i] = a_ori[i];
    if (a_scene[i] < 0 || a_scene[i] >= n_scenes || (i && a_scene[i] < a_scene[i - 1])) return fail(PS_E_ARG, "agent tokens must be scene-major");
    r_agent[a_scene[i] + 1]++;
  }
  int maxA = 1, maxM = 1;
  for (int b = 0; b < n_scenes; ++b) {
    maxA = std::max(maxA, r_agent[b + 1]);
    maxM = std::max(maxM, r_map[b + 1]);
    r_map[b + 1] += r_map[b];
    r_agent[b + 1] += r_agent[b];
  }
  for (int b = 0; b <= n_scenes; ++b) r_agent[b] += Nm;
  {   // the search kernels take begin / end pairs per scene
    std::vector<int> pm(2 * n_scenes), pa(2 * n_scenes);
    for (int b = 0; b < n_scenes; ++b) {
      pm[2 * b] = r_map[b]; pm[2 * b + 1] = r_map[b + 1];
      pa[2 * b] = r_agent[b]; pa[2 * b + 1] = r_agent[b + 1];
    }
    r_map.swap(pm);
    r_agent.swap(pa);
  }
  for (int i = 0; i < A; ++i) {
    if (p_type[i] < 1 || p_type[i] > c.num_agent_types) return fail(PS_E_ARG, "agent_type outside 1..num_agent_types");
    if (p_scene[i] < 0 || p_scene[i] >= n_scenes) return fail(PS_E_ARG, "policy batch_idx out of range");
    pscene[i] = p_scene[i];
  }
  DevBuf<float> d_pos, d_ori, d_atok, d_mtok, d_ppos, d_pori, d_x, d_kva, d_kvm, d_motion, d_traj, d_vel;
  DevBuf<_Float16> d_kha, d_khm;
  DevBuf<int> d_rmap, d_ragent, d_pscene, d_ptype;
  EdgeSet ea, em;
  DevBuf<ChainStep> d_steps;
  auto mn = [](int a, int b) { return a < b ? a : b; };
  const int da = mn(c.pol_max_neigh, maxA), dm = mn(c.pol_max_neigh, maxM);
  const int L = c.pol_layers;
  const int OUT = c.motion_k * c.target_steps * c.state_dim;
  std::vector<int> ptype(p_type, p_type + A);
  if (check_c16_offsets((size_t)Nm + Na, {(size_t)A * da, (size_t)A * dm})) return PS_E_ARG;
  if (c.rel_pos_knn && std::max(maxA, maxM) > 64 * KNN_SLOTS) return fail(PS_E_ARG, "more than 2560 candidate tokens in one scene (knn candidate registers)");
  if (upload(d_pos, pos.data(), pos.size(), st) || upload(d_ori, ori.data(), ori.size(), st) ||
      d_atok.ensure((size_t)std::max(Na, 1) * D) || d_mtok.ensure((size_t)std::max(Nm, 1) * D) ||
      (Na > 0 && upload(d_atok, a_tok, (size_t)Na * D, st)) || (Nm > 0 && upload(d_mtok, m_tok, (size_t)Nm * D, st)) ||
      upload(d_ppos, p_pos, (size_t)A * 2, st) || upload(d_pori, p_ori, (size_t)A, st) || upload(d_x, p_emd, (size_t)A * D, st) ||
      upload(d_rmap, r_map.data(), r_map.size(), st) || upload(d_ragent, r_agent.data(), r_agent.size(), st) ||
      upload(d_pscene, pscene.data(), pscene.size(), st) || upload(d_ptype, ptype.data(), ptype.size(), st) ||
      d_kva.ensure((size_t)L * std::max(Na, 1) * 256) || d_kvm.ensure((size_t)L * std::max(Nm, 1) * 256) ||
      d_kha.ensure((size_t)L * std::max(Na, 1) * 256) || d_khm.ensure((size_t)L * std::max(Nm, 1) * 256) ||
      d_motion.ensure((size_t)A * OUT) || d_traj.ensure((size_t)A * 16 * 4) || d_vel.ensure((size_t)A * 16 * 2) ||
      edge_alloc(ea, A, (size_t)A * da, da, e->stream) || edge_alloc(em, A, (size_t)A * dm, dm, e->stream))
    return fail(PS_E_HIP, "ps_policy_forward: upload/alloc failed");
  // the launch helpers read token geometry from the engine: swap in the caller's arrays for this call
  std::swap(e->d_tok_pos, d_pos);
  launch_kv(e, d_atok.p, Na, e->L_a2p, L, d_kva.p, d_kha.p, (size_t)Na * 256);
  launch_kv(e, d_mtok.p, Nm, e->L_m2p, L, d_kvm.p, d_khm.p, (size_t)Nm * 256);
  const int pe_mode = use_c16(e, A, 2) ? 2 : 1;
  const PeLearnW* la = e->pe_on[2] ? &e->pe_learn[4] : nullptr;
  const PeLearnW* lm = e->pe_on[2] ? &e->pe_learn[5] : nullptr;
  launch_radius(e, ea, d_ragent.p, nullptr, d_ppos.p, d_pscene.p, A, c.pol_agent_radius, c.pol_max_neigh, -1, d_ori.p, d_pori.p, nullptr, 0, pe_mode, la, c.rel_pos_knn != 0);
  launch_radius(e, em, d_rmap.p, nullptr, d_ppos.p, d_pscene.p, A, c.pol_map_radius, c.pol_max_neigh, -1, d_ori.p, d_pori.p, nullptr, 0, pe_mode, lm, c.rel_pos_knn != 0);
  std::vector<ChainStep> hs;
  for (int i = 0; i < L; ++i) {
    ChainStep s1;
    s1.w = e->a2p[i]; s1.kv = d_kva.p + (size_t)i * Na * 256 - (size_t)Nm * 256; s1.eoff = ea.eoff.p; s1.esrc = ea.esrc.p; s1.toff = ea.toff.p; s1.rtT = ea.rtT.p; s1.rtA = ea.rtA.p; s1.kr = la ? 4 : 3; s1.geo = ea.geo.p; s1.khl = d_kha.p + (size_t)i * Na * 256 - (size_t)Nm * 256;
    hs.push_back(s1);
    ChainStep s2;
    s2.w = e->m2p[i]; s2.kv = d_kvm.p + (size_t)i * Nm * 256; s2.eoff = em.eoff.p; s2.esrc = em.esrc.p; s2.toff = em.toff.p; s2.rtT = em.rtT.p; s2.rtA = em.rtA.p; s2.kr = lm ? 4 : 3; s2.geo = em.geo.p; s2.khl = d_khm.p + (size_t)i * Nm * 256;
    hs.push_back(s2);
  }
  int rc = 0;
  if (upload(d_steps, hs.data(), hs.size(), st)) rc = fail(PS_E_HIP, "step upload failed");
  if (!rc) rc = use_c16(e, A, 2) ? launch_chain16(e, d_x.p, A, d_steps.p, 2 * L, false, nullptr, false)
                                      : launch_chain(e, d_x.p, A, 0, 2 * L, std::max(da, dm), false, d_steps.p, 0, la ? 4 : 0);
  if (!rc) {
    // head with a neutral state (last pose = origin, heading 0): only motion_pred is read back
    (void)hipMemsetAsync(d_traj.p, 0, sizeof(float) * (size_t)A * 16 * 4, st);
    std::vector<float> unit((size_t)A * 16 * 4, 0.f);
    for (size_t i = 0; i < (size_t)A * 16; ++i) unit[i * 4 + 3] = 1.f;
    (void)hipMemcpyAsync(d_traj.p, unit.data(), sizeof(float) * unit.size(), hipMemcpyHostToDevice, st);
    const int Gh = c.k_pred_mlp ? 16 : 16 / c.motion_k;
    hipLaunchKernelGGL(k_policy_head_mfma, dim3((A + Gh - 1) / Gh), dim3(256), 0, st, e->head, (const float*)d_x.p, (const int*)d_ptype.p, A,
                       c.motion_k, c.target_steps, c.state_dim, d_motion.p, d_traj.p, d_vel.p, 16, 1, 0, c.ln_eps, (const int*)nullptr,
                       (const float*)nullptr, c.no_pred_vel ? -1 : (c.pred_gmm ? 6 : 3), c.k_pred_mlp ? 1 : 0);
    if (hipStreamSynchronize(st) != hipSuccess) rc = fail(PS_E_HIP, "ps_policy_forward: kernel failure");
  }
  std::swap(e->d_tok_pos, d_pos);
  if (!rc) {
    if (hipMemcpy(motion_pred, d_motion.p, sizeof(float) * (size_t)A * OUT, hipMemcpyDeviceToHost) != hipSuccess ||
        (fused_out && hipMemcpy(fused_out, d_x.p, sizeof(float) * (size_t)A * D, hipMemcpyDeviceToHost) != hipSuccess))
      rc = fail(PS_E_HIP, "ps_policy_forward: D2H failed");
  }
  for (DevBuf<float>* b : {&d_pos, &d_ori, &d_atok, &d_mtok, &d_ppos, &d_pori, &d_x, &d_kva, &d_kvm, &d_motion, &d_traj, &d_vel}) b->release();
  for (DevBuf<int>* b : {&d_rmap, &d_ragent, &d_pscene, &d_ptype}) b->release();
  d_kha.release(); d_khm.release();
  for (EdgeSet* s_ : {&ea, &em}) { s_->cnt.release(); s_->eoff.release(); s_->esrc.release(); s_->edst.release(); s_->toff.release(); s_->tdst.release(); s_->rtA.release(); s_->rtT.release(); s_->geo.release(); }
  d_steps.release();
  return rc;
}

#ifdef PS_C16_DBG
extern "C" int ps_test_c16_dbg(float* out_host, int rows) {   // out_host == nullptr: arm (allocate + zero); else copy the planes back
  static float* d = nullptr;
  static int n = 0;
  if (!out_host) {
    if (d) (void)hipFree(d);
    n = rows;
    if (hipMalloc(&d, (size_t)6 * rows * 128 * 4) != hipSuccess) return -1;
    (void)hipMemset(d, 0, (size_t)6 * rows * 128 * 4);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ps::g_c16_dbg), &d, sizeof(d));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(ps::g_c16_dbg_rows), &n, sizeof(n));
    return (int)hipDeviceSynchronize();
  }
  (void)hipDeviceSynchronize();
  return (int)hipMemcpy(out_host, d, (size_t)6 * n * 128 * 4, hipMemcpyDeviceToHost);
}
#endif
