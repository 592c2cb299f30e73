// Host orchestration of one `fix conp` instance on one MI355X and the C ABI of include/conp_hip.h.
// Method names mirror FixConp / KSpaceModuleEwald (fix_conp.h:37-74, kspacemodule.h:30-40); citations are
// file:line in /root/reference.  There is no CPU fallback: every compute entry point needs the HIP device.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the entry points are resolved with dlopen when a communicator is asked for

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/conp_hip.h"
#include "conp_host.hpp"
#include "conp_kernels.h"

namespace {

thread_local std::string g_last_error;

struct ConpError : std::runtime_error {
  int code;
  ConpError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      throw ConpError(CONP_ERR_NO_DEVICE, std::string("HIP error: ") + hipGetErrorString(e_) + " at " #expr); \
  } while (0)

// CONP_GUARD=1 (diagnostic; tests/test_gpu_guard.py): every device buffer of the library is allocated with a 4-KB zone of a
// known byte pattern before and after it, and conp_debug_check_guards() reads all zones back -- an out-of-range STORE of any kernel
// shows up as a damaged zone with the buffer's size, instead of as silent corruption of a neighbouring allocation or as an abort
// of the HIP runtime somewhere else (round 3 had one abort whose message was lost).  Off: no cost.
struct GuardZones {
  static constexpr size_t G = 4096;
  static constexpr int PATTERN = 0xA5;
  std::mutex m;
  std::map<const void *, size_t> live;       // raw allocation -> payload bytes
  static bool on() { static const bool v = getenv("CONP_GUARD") != nullptr; return v; }
  static GuardZones &get() { static GuardZones g; return g; }
  int bad_freed = 0;                         // damaged zones found when a buffer was released (checked then: the zones go with it)
  std::string what_freed;
  void add(const void *raw, size_t bytes) { std::lock_guard<std::mutex> l(m); live[raw] = bytes; }
  void drop(const void *raw) {
    std::lock_guard<std::mutex> l(m);
    auto it = live.find(raw);
    if (it == live.end()) return;
    (void)hipDeviceSynchronize();
    bad_freed += check_one(static_cast<const char *>(raw), it->second, what_freed, bad_freed);
    live.erase(it);
  }
  static int check_one(const char *raw, size_t bytes, std::string &what, int already) {
    std::vector<unsigned char> h(2 * G);
    if (hipMemcpy(h.data(), raw, G, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(h.data() + G, raw + G + bytes, G, hipMemcpyDeviceToHost) != hipSuccess) { what += " [unreadable zone]"; return 1; }
    int bad = 0;
    for (int side = 0; side < 2; ++side) {
      size_t first = G;
      for (size_t i = 0; i < G; ++i) if (h[side * G + i] != PATTERN) { first = i; break; }
      if (first == G) continue;
      ++bad;
      if (already + bad <= 4) {
        char line[160];
        std::snprintf(line, sizeof line, " [buffer of %zu bytes: zone %s it damaged from byte %zu]", bytes, side ? "behind" : "before", first);
        what += line;
      }
    }
    return bad;
  }
  // number of damaged zones, of live buffers and of buffers released since the process started; `what` names the first few
  int check(std::string &what) {
    std::lock_guard<std::mutex> l(m);
    (void)hipDeviceSynchronize();
    int bad = bad_freed;
    what = what_freed;
    for (const auto &kv : live) bad += check_one(static_cast<const char *>(kv.first), kv.second, what, bad);
    return bad;
  }
};

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  void free_() {
    if (!p) return;
    if (GuardZones::on()) {
      char *raw = reinterpret_cast<char *>(p) - GuardZones::G;
      GuardZones::get().drop(raw);
      (void)hipFree(raw);
    } else (void)hipFree(p);
    p = nullptr;
  }
  ~DevBuf() { free_(); }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  void reserve(size_t count) {
    if (count <= n) return;
    free_();
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (GuardZones::on()) {
      constexpr size_t G = GuardZones::G;
      const size_t padded = (bytes + 15) / 16 * 16;
      char *raw = nullptr;
      HIP_TRY(hipMalloc(reinterpret_cast<void **>(&raw), padded + 2 * G));
      HIP_TRY(hipMemset(raw, GuardZones::PATTERN, G));
      HIP_TRY(hipMemset(raw + G + padded, GuardZones::PATTERN, G));
      GuardZones::get().add(raw, padded);
      p = reinterpret_cast<T *>(raw + G);
    } else HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), bytes));
    n = count;
  }
  void upload(const T *h, size_t count, hipStream_t s) {
    reserve(count);
    if (count) HIP_TRY(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void upload(const std::vector<T> &v, hipStream_t s) { upload(v.data(), v.size(), s); }
  void zero(hipStream_t s) { if (n) HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
  void release() { free_(); n = 0; }
};

// per-kernel timing with HIP events on the library's stream (bench.py's roofline leg)
struct Profiler {
  bool on = false;
  bool dominant_only = false;      // mode 2: events around sk_gemm only (cheap enough to ride inside a timed region)
  bool skipped = false;            // the begin() that belongs to the next end() recorded nothing
  unsigned n_dominant = 0;
  struct Rec { std::string name; hipEvent_t a, b; };
  std::vector<Rec> pending;
  std::vector<hipEvent_t> pool;    // events are reused: creating a pair per kernel per update costs host time
  std::vector<std::string> order;
  std::map<std::string, std::pair<double, int>> acc;
  std::vector<std::string> names_keep;
  hipEvent_t get() {
    hipEvent_t e = nullptr;
    if (!pool.empty()) { e = pool.back(); pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
  }
  void begin(const char *name, hipStream_t s) {
    if (!on) return;
    skipped = dominant_only && std::strcmp(name, "sk_gemm") != 0 && std::strcmp(name, "zn_gemm") != 0;
    // (an event pair drains the queue on both sides of the kernel: ~5 us per update when every launch carries one, measured
    //  0.3057 -> 0.3110 ms at the headline size; every 4th launch keeps the timed region within 0.4 % of an uninstrumented run)
    if (!skipped && dominant_only && (n_dominant++ & 3) != 0) skipped = true;
    if (skipped) return;
    if (pending.size() >= 4096) collect();        // a host that leaves profiling on for a long run must not grow this forever
    Rec r; r.name = name;
    r.a = get(); r.b = get();
    (void)hipEventRecord(r.a, s);
    pending.push_back(r);
  }
  void end(hipStream_t s) { if (on && !skipped && !pending.empty()) (void)hipEventRecord(pending.back().b, s); }
  void collect() {
    for (auto &r : pending) {
      (void)hipEventSynchronize(r.b);
      float ms = 0; (void)hipEventElapsedTime(&ms, r.a, r.b);
      if (!acc.count(r.name)) order.push_back(r.name);
      auto &e = acc[r.name]; e.first += ms; e.second += 1;
      pool.push_back(r.a); pool.push_back(r.b);
    }
    pending.clear();
  }
  void reset() { collect(); acc.clear(); order.clear(); n_dominant = 0; }
  ~Profiler() { for (auto e : pool) (void)hipEventDestroy(e); }
};

// ---- ranks: the host's conp_comm callbacks behind the RankOps interface of conp_host.hpp ----------------------------------
struct RankComm : conp::RankOps {
  conp_comm c{};
  bool have = false;            // callbacks present: the atoms handed to the hooks are this rank's sub-domain
  int rank_ = 0, nranks_ = 1;
  int nranks() const override { return nranks_; }
  int rank() const override { return rank_; }
  bool active() const { return have && nranks_ > 1; }
  static void check(int rc, const char *what) {
    if (rc != 0) throw ConpError(CONP_ERR_STATE, std::string("conp_comm callback failed: ") + what);
  }
  void allreduce_max_int(int *v, int n) override { if (active()) check(c.allreduce_max_int(c.ctx, v, n), "allreduce_max_int"); }
  void allgather_int(int v, int *out) override {
    if (active()) check(c.allgather_int(c.ctx, v, out), "allgather_int"); else out[0] = v;
  }
  void allgatherv_int(const int *send, int n, int *recv, const int *counts, const int *displs) override {
    if (!active()) { for (int i = 0; i < n; ++i) recv[i] = send[i]; return; }
    std::vector<int64_t> cb(nranks_), db(nranks_);
    for (int r = 0; r < nranks_; ++r) { cb[r] = (int64_t)counts[r] * sizeof(int); db[r] = (int64_t)displs[r] * sizeof(int); }
    check(c.allgatherv(c.ctx, send, (int64_t)n * sizeof(int), recv, cb.data(), db.data()), "allgatherv");
  }
  void sum(double *b, int64_t n) { if (active() && n > 0) check(c.allreduce_sum(c.ctx, b, n), "allreduce_sum"); }
  // rank r contributes counts[r] doubles (w values per item); recv holds them rank-major
  void gatherv(const double *send, const std::vector<int> &items, int w, double *recv) {
    if (!active()) { std::memcpy(recv, send, (size_t)items[0] * w * sizeof(double)); return; }
    std::vector<int64_t> cb(nranks_), db(nranks_);
    int64_t off = 0;
    for (int r = 0; r < nranks_; ++r) { cb[r] = (int64_t)items[r] * w * sizeof(double); db[r] = off; off += cb[r]; }
    check(c.allgatherv(c.ctx, send, cb[rank_], recv, cb.data(), db.data()), "allgatherv");
  }
};

// ---- RCCL, resolved at run time (the library stays loadable where no RCCL is installed; torch's bundled copy and the
// system's share a soname, whichever the process loaded first serves both) -------------------------------------------------
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  void load() {
    if (lib) return;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) throw ConpError(CONP_ERR_NO_DEVICE, std::string("cannot load librccl: ") + dlerror());
    auto sym = [&](const char *n) {
      void *p = dlsym(lib, n);
      if (!p) throw ConpError(CONP_ERR_NO_DEVICE, std::string("librccl lacks ") + n);
      return p;
    };
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
  }
  void ok(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw ConpError(CONP_ERR_NO_DEVICE, std::string("RCCL error in ") + what + ": " + (GetErrorString ? GetErrorString(r) : "?"));
  }
};
Rccl g_rccl;   // plain pointers, no destructor: nothing of it runs at exit

}  // namespace

using namespace conp;

struct conp_fix {
  conp_fix_args args{};
  conp_env env{};
  std::vector<double> cutsq_h;
  // host state
  KTables kt;
  KPlan plan;
  PppmPlan pppm;
  PppmDev dpppm{};
  EleIndex idx;
  PairRows brows, arows;
  ListView alist, blist;
  bool have_alist = false, have_blist = false, kspace_ready = false, matrix_loaded = false;
  int runstage = 0;          // fix_conp.cpp:181-183
  int ne_pad = 0, nl = 0, nl_pad = 0, nall = 0;
  int row0 = 0, row1 = 0, rows_per = 0, num_cus = 256;
  // several ranks: host callbacks (spatially decomposed atoms) and / or an RCCL communicator (device-resident collectives)
  RankComm rc;
  bool decomposed = false;       // conp_fix_set_comm was called: atoms and lists are this rank's sub-domain
  ncclComm_t nccl = nullptr;
  bool s_sharded = false;        // the projected inverse is stored by rows [row0, row1) only (d_Srows)
  std::vector<int> elyte_counts;   // decomposed: charged electrolyte atoms per rank (gather layout of d_xg / d_qg)
  int nl_local = 0;
  std::vector<double> xq_pack, xq_all;
  std::vector<SkItem> items_h;   // sk_gemm work items of this rank
  std::vector<SkTile> tiles_h;   // (row tile, col tile) pairs of this rank, sorted by col tile
  std::vector<int> ele_pairs_h;    // (atom index, eleall index) of every owned or ghost electrode atom
  int n_ele_atoms = 0;
  std::vector<double> tile_flo, tile_fhi;   // this rank's share of each of its tiles, as fractions of the tile's chunk axis
  std::vector<int> ct_ptr_h, seg_ptr_h, seg_idx_h, own_rt_h;   // own_rt_h: the row tiles this rank works on (sorted)
  double evscale = 0, totsetq = 0, scalar_output = 0, totinve = 0, slabcorr = 0;
  int cg_iterations = 0;
  int inverse_path = 0;          // conp_info.inverse_path: 1 positive-definite elimination, 2 partial pivoting
  int inverse_retries = 0;       // times the last inverse fell back from the multi-workgroup panel (barrier time-out, info = -7)
  // the fix's log file (fix_conp.cpp:119): lines are buffered here and handed to the host by conp_fix_log_drain
  std::string logbuf, logdrain, mesgbuf, mesgdrain;   // mesgbuf: what the reference sends to utils::logmesg (:460, :1008)
  // CONP_TIME_HOST=1: where a host-buffer update spends its host time (printed to stderr when the handle is closed)
  const bool time_host = getenv("CONP_TIME_HOST") != nullptr;
  double th[6] = {0, 0, 0, 0, 0, 0};    // list check, staging copies, enqueue, wait, scatter, updates
  static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  bool pp_im_clean = false;      // the mesh's imaginary brick is all zero (left so by the b path's last backward pass)
  // PPPMCONP's per-step caches (pppm_conp.cpp: elyte_density_brick + elyte_mapped, u_brick): pp_keep -- a host that overrides
  // PPPM::make_rho asked for the electrolyte brick of every b_cal (conp_pppm_keep_density); pp_elyte_valid -- d_pp_elyte holds the
  // brick of the LAST b_cal and nothing has moved since; pp_u_valid -- d_pp_re holds the mesh potential of the total density
  // (conp_pppm_compute and the other collective entries leave it), so the per-atom entry only gathers from it.
  bool pp_keep = false, pp_elyte_valid = false, pp_u_valid = false;
  int pp_elyte_spreads = 0;      // how often the electrolyte atoms were spread onto the mesh (b_cal and density queries)
  DevBuf<double> d_pp_elyte, d_pp_xg, d_pp_qg;
  DevBuf<int> d_pp_iota;
  double Btime = 0., Ctime = 0., Ktime = 0.;      // accumulated like :549-552 (seconds)
  hipEvent_t ev_b[3] = {nullptr, nullptr, nullptr};
  bool ev_pending = false;
  int nzc = 0;               // distinct electrode z values (<= 64: planar fast path of the projection), else 0
  std::vector<double> csk_h, snk_h, xele_h, d_vec_h;
  // EHGO pair mode (fix_conp.cpp:1482-1598)
  bool ehgo_active = false;
  double kappa = 1.0;
  std::vector<double> eta_i_h, u0_i_h, eta_ij_h, fo_ij_h;
  std::string warning;
  double cond_vmult = 0.0;       // fix cond (fix_cond.cpp:58-68)
  bool cond_ready = false;
  std::vector<int> elyte_idx_h;
  int bl_inum = 0;                 // owners in the ele-electrolyte list as uploaded for the post-force kernel
  size_t bl_nneigh = 0;            // length of its flattened neighbour array
  DevBuf<char> d_rows_scratch;
  int nlocal_cur = 0;
  // device state
  hipStream_t stream = nullptr;
  bool own_stream = false;
  DevBuf<double> d_x, d_q, d_qc, d_slab_part, d_Gpart, d_G, d_Gw, d_wfull, d_Rp, d_Tz, d_ele_z, d_bk, d_breal, d_b_own,
      d_eleallq_own, d_qele, d_elesetq, d_eleinitq, d_A, d_cutsq, d_scalars, d_ainve, d_sfr, d_sfi, d_cg_res, d_cg_p, d_Srows, d_xg, d_qg, d_pp_ele, d_pp_scratch,
      d_cg_ap, d_cg_scal, d_inv_work, d_inv_backup, d_Tzc, d_TzcT, d_Hpart, d_Hc, d_Wz, d_Spk, d_yp, d_f, d_pfacc, d_pp_coeff, d_pp_green, d_pp_tw0, d_pp_tw1,
      d_pp_tw2, d_pp_re, d_pp_im, d_pp_ew, d_eta_ij, d_fo_ij, d_u0_i, d_diag_atom, d_setzvec;
  DevBuf<double2> d_Xt, d_Yt, d_Zt, d_Xe, d_Ye;      // d_Xe / d_Ye: electrode atoms' axis phases [k][ne_pad] (once per run)
  DevBuf<int> d_type, d_atom2eleall, d_elyte_idx, d_p_ikx, d_p_iky, d_p_sgn, d_sf_row_a, d_sf_col_c, d_k_sign, d_k_p, d_k_m,
      d_elecheck, d_zclass, d_nb_act, d_rt_mine, d_own_rt, d_own_pv, d_ele_pairs, d_a_chunk_group, d_ct_ptr, d_seg_ptr, d_seg_idx, d_hslot_ptr, d_hslot_idx, d_b_rowptr, d_b_ele, d_b_oth, d_a_rowptr, d_a_ele, d_a_oth, d_a_col, d_bl_ilist, d_bl_numneigh, d_bl_first, d_bl_neigh, d_pp_egrid, d_ipiv, d_info, d_cg_done, d_iota, d_ele_csr_ptr, d_ele_csr_of, d_ele_csr_row;
  bool left_stale = false;          // the fused GEMV + charge write leaves the fix scalar's group-1 sum to refresh_scalar()
  double left_potdiff = 0.0;
  DevBuf<unsigned char> d_mask;
  DevBuf<SkItem> d_items;
  DevBuf<SkWItem> d_witems;
  int witems_maxseg = 1;
  DevBuf<SkProj> d_skproj;
  SkProj skproj_h{};
  DevBuf<SkFuse> d_skfuse;
  SkFuse skfuse_h{};
  DevBuf<SkTile> d_tiles;
  double *d_b = nullptr, *d_eleallq = nullptr;   // bound (external) or own buffers
  bool b_bound = false, q_bound = false;         // conp_fix_bind_device_buffers gave us the host's vectors
  int n_slab_part = 0;
  const bool no_fuse = diag_switch("CONP_NO_FUSE") != nullptr;   // experiment switch: separate sk_reduce / b_hc launches
  const bool shard_by_cost = diag_switch("CONP_SHARD_COST_AXIS") != nullptr;   // comparison switch: rank shards = shares of the cost axis (round 2)
  const bool sk_partials = path_on(CONP_PATH_PARTIAL_TILES);   // comparison switch: partial tiles + reducing launch, no projection in sk_gemm
  const char *hc_presum_env = diag_switch("CONP_HC_PRESUM");      // comparison switch: 1 / 0 = always / never add the pieces in a launch of their own
  // The host-buffer hooks report Ktime / Ctime (fix_conp.cpp:553-568).  By default they run the SAME kernels as the device hooks
  // (bitwise-equal charges): the pair sums share a launch with the k-space phases, so Ctime stays 0 and Ktime holds all of b_cal.
  // CONP_TIME_SPLIT=1 launches the two halves separately (last-ulp different dot order) so that each gets its own figure.
  const bool time_split = path_on(CONP_PATH_TIME_SPLIT);
  // the dot workgroups' bounded wait for the class table (b_zc_fused_kernel); CONP_PATH_HC_NO_WAIT: zero -- every dot workgroup adds
  // the pieces itself, the path a wait that runs out takes (tested against the hand-off: the same bits)
  const unsigned hc_spin_limit = path_on(CONP_PATH_HC_NO_WAIT) ? 0u : (1u << 16);
  // Measured and NOT the default (round 5, tools/ab_libs.sh on one box, profiles/r05_tail_handoff_ab.txt): the pieces' sums, the pair
  // sums and the dot as ONE launch with the fence-free hand-off -- reduce_project 24.7-25.3 us against 19.0-19.1 for hc_sum_kernel +
  // b_zc_final_kernel (headline), 26.5-27.0 against 20.0-20.4 (slab geometry).  What a kernel boundary does in ~2 us (make 18 KB
  // visible to 256 workgroups on eight XCDs) costs the hand-off a write-through, a ticket, a poll and 256 x 18 KB of sc1 reads from
  // the memory side; the pair rows, which filled the idle CUs beside hc_sum's handful of workgroups, sit on every dot workgroup's
  // own path instead.  Kept as a test path (bit-identical b, tests/test_gpu_decks.py).
  const bool hc_fused = path_on(CONP_PATH_HC_FUSED);
  const bool gemv_rows = path_on(CONP_PATH_GEMV_ROWS);      // test path: the row-by-row product at every size
  const bool no_phase_fuse = path_on(CONP_PATH_PHASE_LAUNCH);      // comparison switch: always the stand-alone phase launch
  const bool no_ride = diag_switch("CONP_NO_RIDE") != nullptr;      // comparison switch: the real-space pair sums in a launch of their own (b_real_combine)
  int table_c0 = 0, table_c1 = 0;  // chunk range (16 atoms each) whose phase tables this rank's sk_gemm reads
  int hslots = 0;                  // entries of the owned row tiles' segment lists (d_hslot_idx)
  bool g_current = true;           // d_G holds the last update's structure factors (false after a projecting update: conp_fix_get_sfac re-forms it)
  int max_nsplit = 0;              // most output slots any tile of the plan has (chooses sk_reduce's one- or two-level sum)
  int n_slots = 0;                 // output slots of sk_gemm: one per (segment, row tile of the plan its band touches)
  DevPlan dplan{};
  Profiler prof;

  ~conp_fix() {
    if (time_host && th[5] > 0)
      std::fprintf(stderr, "conp host-buffer update, us per call over %.0f calls: list check %.1f, staging + H2D enqueue %.1f, kernel "
                           "enqueue %.1f, solve enqueue + wait + D2H %.1f, scatter %.1f\n", th[5], 1e6 * th[0] / th[5], 1e6 * th[1] / th[5],
                   1e6 * th[2] / th[5], 1e6 * th[3] / th[5], 1e6 * th[4] / th[5]);
    prof.collect();
    drop_graph();
    if (nccl) { (void)hipStreamSynchronize(stream); (void)g_rccl.CommDestroy(nccl); nccl = nullptr; }
    if (pin_x || pin_q) { (void)hipStreamSynchronize(stream); if (pin_x) (void)hipHostUnregister(const_cast<double *>(pin_x)); if (pin_q) (void)hipHostUnregister(const_cast<double *>(pin_q)); }
    if (h_pin) { (void)hipStreamSynchronize(stream); (void)hipHostFree(h_pin); }
    if (h_np) { (void)hipStreamSynchronize(stream); (void)hipHostFree(h_np); }
    if (ren_arena) { (void)hipStreamSynchronize(stream); (void)hipHostFree(ren_arena); ren_arena = nullptr; }
    if (zn_flag_host) { (void)hipStreamSynchronize(stream); (void)hipHostFree(zn_flag_host); zn_flag_host = nullptr; }
    for (auto &e : ev_b) if (e) (void)hipEventDestroy(e);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }

  RealParams real_params() const {
    RealParams rp;
    rp.g_ewald = env.g_ewald; rp.eta = args.eta;
    double cut_coulsq = env.cut_coul * env.cut_coul;                  // fix_conp.cpp:1237-1240
    const double cut_erfc = 5.8 * 5.8 / (env.g_ewald * env.g_ewald);
    if (cut_coulsq > cut_erfc) cut_coulsq = cut_erfc;
    rp.cut_coulsq = cut_coulsq; rp.ntypes = env.ntypes; rp.cutsq = d_cutsq.p;
    rp.ehgo = ehgo_active ? 1 : 0; rp.eta_ij = d_eta_ij.p; rp.fo_ij = d_fo_ij.p; rp.u0_i = d_u0_i.p;
    return rp;
  }

  // The wait at the end of a host-buffer update: the runtime's blocking synchronisation wakes the thread tens of microseconds after
  // the stream drained; polling the stream for the first two milliseconds (an update takes 0.03 - 20 ms) returns within a few.
  // Longer waits (setup, large systems) fall back to the blocking call: no core is burnt for them.
  const bool sync_block = diag_switch("CONP_SYNC_BLOCK") != nullptr;      // comparison switch: always the blocking call
  const bool results_by_copy = diag_switch("CONP_RESULTS_COPY") != nullptr;      // comparison switch: charges / scalars back by hipMemcpyAsync
  void sync() {
    if (!sync_block) {
      const double t0 = now_s();
      for (;;) {
        const hipError_t e = hipStreamQuery(stream);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) HIP_TRY(e);
        if (now_s() - t0 > 2e-3) break;
      }
    }
    HIP_TRY(hipStreamSynchronize(stream));
  }

  // ---------------------------------------------------------------------------------------------
  void init_device() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw ConpError(CONP_ERR_NO_DEVICE, "no HIP device visible: libconp_hip has no CPU fallback");
    // "this rank's GPU": -1 = global rank modulo the visible devices; -(2 + l) = NODE-LOCAL rank l modulo the visible devices (what
    // an MPI host passes: with several nodes the global rank says nothing about which of THIS node's GPUs is free)
    if (env.device <= -2) env.device = (-env.device - 2) % ndev;
    else if (env.device < 0) env.device = env.rank % ndev;
    if (env.device >= ndev) throw ConpError(CONP_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(env.device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, env.device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
      throw ConpError(CONP_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    num_cus = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreate(&stream));
    own_stream = true;
    d_cutsq.upload(cutsq_h, stream);
    d_scalars.reserve(16);
    d_scalars.zero(stream);

  }

  // FixConp::modify_param (fix_conp.cpp:1482-1515)
  int modify_param(int narg, const char *const *arg) {
    if (!args.ehgo) throw ConpError(CONP_ERR_ARG, "Can't fix_modify conp parameters in basic pair mode");
    if (narg < 1 || std::strcmp(arg[0], "ehgo") != 0) return 0;
    const int nt1 = env.ntypes + 1;
    if (eta_i_h.empty()) { eta_i_h.assign(nt1, 0.0); u0_i_h.assign(nt1, 0.0); }       // ehgo_allocate :1575-1588
    const double CON_s2overPIS = std::sqrt(2.0) / 1.77245385090551602729;
    const double evs = env.qe2f / env.qqr2e;
    if (narg >= 2 && std::strcmp(arg[1], "kappa") == 0) {
      if (narg != 3) throw ConpError(CONP_ERR_ARG, "Invalid number of inputs for EHGO coeff setting");
      kappa = std::atof(arg[2]);
      return 3;
    }
    if (narg >= 2 && std::strcmp(arg[1], "coeff") == 0) {
      if (narg != 5) throw ConpError(CONP_ERR_ARG, "Invalid number of inputs for EHGO coeff setting");
      // utils::bounds(arg[2], 1, ntypes): "N", "*", "N*", "*M", "N*M"
      int ilo = 1, ihi = env.ntypes;
      const std::string b = arg[2];
      const size_t star = b.find('*');
      if (star == std::string::npos) ilo = ihi = std::atoi(b.c_str());
      else {
        if (star > 0) ilo = std::atoi(b.substr(0, star).c_str());
        if (star + 1 < b.size()) ihi = std::atoi(b.substr(star + 1).c_str());
      }
      if (ilo < 1 || ihi > env.ntypes) throw ConpError(CONP_ERR_ARG, "Numeric index is out of bounds");
      const double eta_one = std::atof(arg[3]);
      const double u0_one = std::strcmp(arg[4], "auto") == 0 ? CON_s2overPIS * eta_one / evs : std::atof(arg[4]);
      int count = 0;
      for (int i = ilo; i <= ihi; ++i) { eta_i_h[i] = eta_one; u0_i_h[i] = u0_one * evs; ++count; }   // eV/e^2 -> 1/A :1506
      if (count == 0) throw ConpError(CONP_ERR_ARG, "Couldn't set EHGO coeffs with mintype more than maxtype");
      return 5;
    }
    throw ConpError(CONP_ERR_ARG, "Invalid entry for EHGO coeff setting");
  }

  // ehgo_setup_tables (fix_conp.cpp:1517-1559), called from init() in the reference = before the first setup here
  void ehgo_setup_tables() {
    if (!args.ehgo) return;
    const int nt1 = env.ntypes + 1;
    if (eta_i_h.empty()) { eta_i_h.assign(nt1, 0.0); u0_i_h.assign(nt1, 0.0); }
    bool setflag = false;
    for (int i = 1; i < nt1; ++i) if (eta_i_h[i] || u0_i_h[i]) setflag = true;
    if (!setflag) {       // :1553-1558
      ehgo_active = false;
      warning = "No EHGO settings found, switching back to ETA mode";
      return;
    }
    const double CON_s2overPIS = std::sqrt(2.0) / 1.77245385090551602729, sq8 = std::sqrt(8.0);
    std::vector<double> f_i(nt1, 0.0);
    eta_ij_h.assign((size_t)nt1 * nt1, 0.0); fo_ij_h.assign((size_t)nt1 * nt1, 0.0);
    for (int i = 1; i < nt1; ++i) f_i[i] = u0_i_h[i] - CON_s2overPIS * eta_i_h[i];
    for (int i = 1; i < nt1; ++i)
      for (int j = 1; j <= i; ++j) {
        if (eta_i_h[i] && eta_i_h[j]) {
          const double etasq = eta_i_h[i] * eta_i_h[i] + eta_i_h[j] * eta_i_h[j];
          const double etaprod = eta_i_h[i] * eta_i_h[j];
          const double eij = etaprod / std::sqrt(etasq);
          const double o_ij = sq8 * eij * eij * eij / (etaprod * std::sqrt(etaprod));
          const double f_ij = 0.5 * kappa * (f_i[i] + f_i[j]);
          eta_ij_h[(size_t)i * nt1 + j] = eij;
          fo_ij_h[(size_t)i * nt1 + j] = f_ij * o_ij;
        } else eta_ij_h[(size_t)i * nt1 + j] = eta_i_h[i] + eta_i_h[j];
        if (i != j) { eta_ij_h[(size_t)j * nt1 + i] = eta_ij_h[(size_t)i * nt1 + j]; fo_ij_h[(size_t)j * nt1 + i] = fo_ij_h[(size_t)i * nt1 + j]; }
      }
    d_eta_ij.upload(eta_ij_h, stream); d_fo_ij.upload(fo_ij_h, stream); d_u0_i.upload(u0_i_h, stream);
    ehgo_active = true;
  }

  // km_ewald.cpp:63-132 conp_setup
  void km_conp_setup(double qsqsum, int64_t natoms) {
    kt.build(env.g_ewald, env.accuracy, env.slab_volfactor, env.slabflag, env.xprd, env.yprd, env.zprd, qsqsum, natoms,
             env.qqrd2e, env.dielectric);
    plan.build(kt);
    ++plan_gen;
    // padding planar rows point at the all-zero X row kxmax+1 (they then contribute nothing)
    // (96 more than the plan's row tiles hold: a band of sk_gemm may run up to five row fragments past the last planar vector)
    std::vector<int> ikx(plan.n_row_tiles * 64 + 96, plan.kxmax + 1), iky(plan.n_row_tiles * 64 + 96, 0), sgn(plan.n_row_tiles * 64 + 96, 1);
    for (int p = 0; p < plan.np; ++p) { ikx[p] = plan.p_ikx[p]; iky[p] = plan.p_iky[p]; sgn[p] = plan.p_sgn[p]; }
    d_p_ikx.upload(ikx, stream); d_p_iky.upload(iky, stream); d_p_sgn.upload(sgn, stream);
    d_wfull.upload(plan.wfull, stream);
    d_sf_row_a.upload(plan.sf_row_a, stream); d_sf_col_c.upload(plan.sf_col_c, stream);
    d_k_sign.upload(plan.k_sign, stream); d_k_p.upload(plan.k_p, stream); d_k_m.upload(plan.k_m, stream);
    d_nb_act.upload(plan.nba_rc, stream);
    dplan = DevPlan{plan.np, plan.nz, plan.n_row_tiles, plan.n_col_tiles, plan.R_pad, plan.C_pad,
                    plan.kxmax, plan.kymax, d_p_ikx.p, d_p_iky.p, d_p_sgn.p, d_nb_act.p, d_wfull.p};
    d_G.reserve((size_t)plan.R_pad * plan.C_pad); d_Gw.reserve((size_t)plan.R_pad * plan.C_pad);
    d_G.zero(stream); d_Gw.zero(stream);
    // k-shard.  The structure-factor work is laid out on ONE axis -- tile after tile (col tile major), chunk after chunk, the axis
    // build_items cuts into workgroup shares -- and rank r takes the r-th N-th of it by cost.  A tile that straddles a rank
    // boundary is shared: each side multiplies its own chunk range, reduces and projects its partial G (the projection is linear
    // in G), and the all-reduce of b adds the pieces.  Every rank gets the same work whatever the number of tiles (the first
    // version dealt whole row tiles, heaviest first: 8 unequal tiles on 8 ranks left the heaviest rank 18 % above the mean, and
    // more ranks than tiles idle).  Boundaries are fractions of a tile here; build_items turns them into chunk numbers -- both
    // neighbours evaluate the same expression, so their ranges meet exactly.  CONP_SHARD_TILES=1: whole tiles (comparison).
    {
      struct GT { int rt, ct; double c; };
      std::vector<GT> all;
      for (int ct = 0; ct < plan.n_col_tiles; ++ct)
        for (int rt = 0; rt < plan.n_row_tiles; ++rt) {
          if (plan.nba(rt, ct) <= 0) continue;
          const unsigned nbf = plan.nfa16(rt, ct);
          double sum = 0.0;
          for (int f = 0; f < 4; ++f) sum += (double)((nbf >> (8 * f)) & 255u);
          all.push_back(GT{rt, ct, 0.125 * sum + SK_C0});     // nbf counts 8-kz column fragments: two per kz block
        }
      std::vector<double> flo(all.size(), 0.0), fhi(all.size(), 0.0);
#ifdef CONP_DIAG
      const bool whole_tiles = getenv("CONP_SHARD_TILES") && env.nranks > 1;     // diagnostic library only
#else
      const bool whole_tiles = false;
#endif
      if (whole_tiles) {
        // whole row tiles, heaviest first to the least loaded rank (every rank computes the same map)
        std::vector<std::pair<double, int>> order;
        std::vector<double> rtc(plan.n_row_tiles, 0.0);
        for (const auto &g : all) rtc[g.rt] += g.c;
        for (int rt = 0; rt < plan.n_row_tiles; ++rt) order.push_back({-rtc[rt], rt});
        std::stable_sort(order.begin(), order.end());
        std::vector<double> load(env.nranks, 0.0);
        std::vector<int> owner(plan.n_row_tiles, 0);
        for (auto &e : order) {
          int best = 0;
          for (int r = 1; r < env.nranks; ++r) if (load[r] < load[best]) best = r;
          owner[e.second] = best;
          load[best] += -e.first;
        }
        for (size_t t = 0; t < all.size(); ++t) if (owner[all[t].rt] == env.rank) fhi[t] = 1.0;
      } else if (!shard_by_cost) {
        // Round 3: rank r takes the r-th N-th of the ATOMS of every tile (the axis the XCD-aware shares cut inside a rank, one level
        // up).  Every rank has every tile with the same chunk count: equal work by construction, whatever the tiles' costs, and a
        // rank needs the phase tables of ITS atoms only -- the phase kernel, replicated until now (12 us of an 89-us update on 8
        // emulated ranks), shrinks with the share.  Each rank projects its partial structure factors for all rows; the
        // all-reduce of b adds them, as it did for tiles that straddled a rank boundary.
        for (size_t t = 0; t < all.size(); ++t) {
          flo[t] = (double)env.rank / (double)env.nranks;
          fhi[t] = (double)(env.rank + 1) / (double)env.nranks;
        }
      } else {
        double W = 0.0;
        for (const auto &g : all) W += g.c;
        const double lo = W * (double)env.rank / (double)env.nranks, hi = W * (double)(env.rank + 1) / (double)env.nranks;
        double S = 0.0;
        for (size_t t = 0; t < all.size(); ++t) {
          const double c = all[t].c;
          flo[t] = std::min(1.0, std::max(0.0, (lo - S) / c));
          fhi[t] = env.rank + 1 == env.nranks && t + 1 == all.size() ? 1.0 : std::min(1.0, std::max(0.0, (hi - S) / c));
          S += c;
        }
      }
      std::vector<int> mine(plan.n_row_tiles, 0);
      tiles_h.clear(); tile_flo.clear(); tile_fhi.clear();
      ct_ptr_h.assign(plan.n_col_tiles + 1, 0);
      for (size_t t = 0; t < all.size(); ++t) {
        if (fhi[t] > flo[t]) {
          const int rt = all[t].rt, ct = all[t].ct;
          tiles_h.push_back(SkTile{rt, ct, plan.nba(rt, ct), 0, 0, plan.nfa16(rt, ct)});
          tile_flo.push_back(flo[t]); tile_fhi.push_back(fhi[t]);
          mine[rt] = 1;
        }
        ct_ptr_h[all[t].ct + 1] = (int)tiles_h.size();
      }
      for (int ct = 0; ct < plan.n_col_tiles; ++ct) ct_ptr_h[ct + 1] = std::max(ct_ptr_h[ct + 1], ct_ptr_h[ct]);
      own_rt_h.clear();
      for (int rt = 0; rt < plan.n_row_tiles; ++rt) if (mine[rt]) own_rt_h.push_back(rt);
      d_rt_mine.upload(mine, stream);
      std::vector<int> own_up = own_rt_h;
      if (own_up.empty()) own_up.push_back(0);
      d_own_rt.upload(own_up, stream);
      // planar vectors of this rank's row tiles, packed for the planar fast path of the projection (b_zc_final_kernel):
      // |kx| in bits 0-11, |ky| in bits 12-23, bit 24 = negative ky; padding vectors: the all-zero X row kxmax + 1
      {
        std::vector<int> pack((size_t)own_up.size() * 64);
        for (size_t u = 0; u < own_up.size(); ++u)
          for (int w = 0; w < 64; ++w) {
            const int p = own_up[u] * 64 + w;
            const int ikx = p < plan.np ? plan.p_ikx[p] : plan.kxmax + 1, iky = p < plan.np ? plan.p_iky[p] : 0;
            const int neg = p < plan.np && plan.p_sgn[p] < 0 ? 1 : 0;
            pack[u * 64 + w] = ikx | (iky << 12) | (neg << 24);
          }
        d_own_pv.upload(pack, stream);
      }
    }
    d_ct_ptr.upload(ct_ptr_h, stream);
    sync();
    kspace_ready = true;
    // the electrode tables, the z-class tables and sk_gemm's projection block (d_skproj: raw pointers into d_wfull / d_TzcT, the
    // plan's C_pad) belong to the plan that was just replaced: whoever needs them next rebuilds them (km_a_read)
    tables_current = false;
  }

  // fix_conp.cpp:393-424 linalg_init
  void linalg_init(const conp_atoms *at) {
    if (runstage != 0 || idx.initialised) return;
    if (args.pppm) {
      // fix_conp.cpp:401-404: the `pppm` keyword needs the pppm/conp kspace style; here: its mesh and order via conp_env
      if (env.pppm_nx <= 0 || env.pppm_ny <= 0 || env.pppm_nz <= 0 || env.pppm_order <= 0)
        throw ConpError(CONP_ERR_ARG, "Fix conp couldn't detect a pppm/conp kspace style (which is required with the pppm flag)");
      const double lo[3] = {env.boxlo_x, env.boxlo_y, env.boxlo_z}, prd[3] = {env.xprd, env.yprd, env.zprd};
      pppm.build(env.pppm_nx, env.pppm_ny, env.pppm_nz, env.pppm_order, env.g_ewald, env.slab_volfactor, lo, prd);
      d_pp_coeff.upload(pppm.rho_coeff, stream); d_pp_green.upload(pppm.greensfn, stream);
      d_pp_tw0.upload(pppm.twid[0], stream); d_pp_tw1.upload(pppm.twid[1], stream); d_pp_tw2.upload(pppm.twid[2], stream);
      d_pp_re.reserve(pppm.nfft); d_pp_im.reserve(pppm.nfft); pp_im_clean = false;
      dpppm.nx = pppm.nx; dpppm.ny = pppm.ny; dpppm.nz = pppm.nz; dpppm.order = pppm.order; dpppm.nlower = pppm.nlower;
      dpppm.nfft = pppm.nfft; dpppm.shift = pppm.shift; dpppm.shiftone = pppm.shiftone; dpppm.delvolinv = pppm.delvolinv;
      for (int c = 0; c < 3; ++c) { dpppm.delinv[c] = pppm.delinv[c]; dpppm.boxlo[c] = pppm.boxlo[c]; }
      dpppm.rho_coeff = d_pp_coeff.p; dpppm.greensfn = d_pp_green.p;
      dpppm.twid[0] = d_pp_tw0.p; dpppm.twid[1] = d_pp_tw1.p; dpppm.twid[2] = d_pp_tw2.p;
    }
    double qn[2] = {0.0, (double)at->nlocal};              // km_ewald.cpp:72-78: sum q^2 over the owned atoms, Allreduce :77
    for (int i = 0; i < at->nlocal; i++) qn[0] += at->q[i] * at->q[i];
    rc.sum(qn, 2);                                         // (natoms = atom->natoms: the owned atoms of all ranks)
    km_conp_setup(qn[0], (int64_t)std::llround(qn[1]));
    evscale = env.qe2f / env.qqr2e;                        // :412
    ehgo_setup_tables();                                   // FixConp::init :296-299
    idx.linalg_init(at->nlocal, at->tag, &rc);
  }

  // ---- uploads of a re-neighbour.  A hipMemcpyAsync out of pageable memory blocks the caller per call (the runtime stages it and
  // waits); a re-neighbour makes a dozen of them.  They go through ONE page-locked arena instead: a memcpy into the arena, a real
  // asynchronous transfer out of it, the host moves on to the next table; the arena is recycled at the next re-neighbour (this one
  // ends with a stream synchronisation).  Arrays the host has page-locked itself (the glue's flattened lists; atom arrays
  // between conp_fix_pin_host_arrays and _unpin_host_arrays) are transferred from where they lie.
  char *ren_arena = nullptr;
  size_t ren_cap = 0, ren_off = 0;
  void ren_begin(size_t bytes_wanted) {
    ren_off = 0;
    if (bytes_wanted <= ren_cap) return;
    if (ren_arena) { sync(); (void)hipHostFree(ren_arena); ren_arena = nullptr; ren_cap = 0; }
    const size_t want = bytes_wanted + bytes_wanted / 4 + 4096;
    if (hipHostMalloc(reinterpret_cast<void **>(&ren_arena), want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ren_arena = nullptr; return; }
    ren_cap = want;
  }
  static bool host_page_locked(const void *h) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, h) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
  }
  template <class T>
  void ren_upload(DevBuf<T> &d, const T *h, size_t count, bool locked = false) {
    d.reserve(count);
    if (!count) return;
    const size_t bytes = count * sizeof(T), o = (ren_off + 63) & ~(size_t)63;
    if (locked) { HIP_TRY(hipMemcpyAsync(d.p, h, bytes, hipMemcpyHostToDevice, stream)); return; }
    if (o + bytes > ren_cap) {
      // the arena is full (first use, or a list that grew): straight out of the caller's pageable memory, and waited for -- the
      // source may be a temporary of the caller
      HIP_TRY(hipMemcpyAsync(d.p, h, bytes, hipMemcpyHostToDevice, stream));
      sync();
      return;
    }
    std::memcpy(ren_arena + o, h, bytes);
    ren_off = o + bytes;
    HIP_TRY(hipMemcpyAsync(d.p, ren_arena + o, bytes, hipMemcpyHostToDevice, stream));
  }
  template <class T>
  void ren_upload(DevBuf<T> &d, const std::vector<T> &v) { ren_upload(d, v.data(), v.size()); }

  void upload_atoms_static(const conp_atoms *at) {
    nall = at->nlocal + at->nghost;
    ren_upload(d_type, at->type, (size_t)nall);
    // one pass: the (atom, eleall) pairs of every owned / ghost electrode atom -- the charge scatter list, and what the device
    // fills its atom -> eleall table from (launch_atom2eleall below)
    ele_pairs_h.clear();
    for (int i = 0; i < nall; ++i) {
      if (!at->echeck[i]) continue;
      const int e = (at->tag[i] <= idx.maxtag_all) ? idx.tag2eleall[at->tag[i]] : -1;
      if (e >= 0 && e < idx.elenum_all) { ele_pairs_h.push_back(i); ele_pairs_h.push_back(e); }
    }
    n_ele_atoms = (int)(ele_pairs_h.size() / 2);
    {   // the same list by electrode row (CSR): the fused GEMV + charge write scatters row by row
      const int ne = idx.elenum_all;
      std::vector<int> ptr((size_t)ne + 1, 0), of(std::max(n_ele_atoms, 1), 0);
      for (int k = 0; k < n_ele_atoms; ++k) ++ptr[ele_pairs_h[2 * (size_t)k + 1] + 1];
      for (int r = 0; r < ne; ++r) ptr[r + 1] += ptr[r];
      std::vector<int> fill(ptr.begin(), ptr.end() - 1);
      for (int k = 0; k < n_ele_atoms; ++k) of[fill[ele_pairs_h[2 * (size_t)k + 1]]++] = ele_pairs_h[2 * (size_t)k];
      std::vector<int> rowof(of.size(), 0);
      for (int r = 0; r < ne; ++r) for (int k = ptr[r]; k < ptr[r + 1]; ++k) rowof[k] = r;
      ren_upload(d_ele_csr_ptr, ptr); ren_upload(d_ele_csr_of, of); ren_upload(d_ele_csr_row, rowof);
    }
    if (ele_pairs_h.empty()) { ele_pairs_h.push_back(0); ele_pairs_h.push_back(0); }
    ren_upload(d_ele_pairs, ele_pairs_h);
    d_atom2eleall.reserve((size_t)std::max(nall, 1));
    launch_atom2eleall(stream, nall, n_ele_atoms, d_ele_pairs.p, d_atom2eleall.p);
    d_x.reserve((size_t)nall * 3); d_q.reserve(nall);
  }

  // real-space rows of b (blist_coul_cal membership, fix_conp.cpp:1326-1350) from the list that is already on the device;
  // CONP_ROWS_HOST=1 keeps the host counting sort (same output) for comparison
  int64_t n_b_pairs = 0;
  unsigned *h_np = nullptr;            // page-locked word the regrouping's pair count lands in
  bool np_pending = false;
  void build_b_rows_device(const conp_atoms *at) {
    const int ne = idx.elenum_all;
    const bool on_host = path_on(CONP_PATH_ROWS_HOST);      // read per call: the A/B test flips it between two handles
    np_pending = false;
    if (on_host) {
      build_b_rows(blist, at->nlocal, at->tag, at->echeck, idx, env.newton_pair != 0, brows);
      d_b_rowptr.upload(brows.row_ptr, stream); d_b_ele.upload(brows.ele_atom, stream); d_b_oth.upload(brows.oth_atom, stream);
      n_b_pairs = brows.npairs();
      return;
    }
    const size_t sb = b_rows_scratch_bytes(ne, bl_nneigh);
    d_rows_scratch.reserve(sb);
    d_b_rowptr.reserve((size_t)ne + 1); d_b_ele.reserve(std::max<size_t>(bl_nneigh, 1)); d_b_oth.reserve(std::max<size_t>(bl_nneigh, 1));
    if (!h_np) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_np), 64, hipHostMallocDefault));
    launch_build_b_rows(stream, bl_inum, bl_nneigh, d_bl_ilist.p, d_bl_numneigh.p, d_bl_first.p, d_bl_neigh.p, d_atom2eleall.p,
                        at->nlocal, env.newton_pair != 0, ne, d_rows_scratch.p, sb, d_b_rowptr.p, d_b_ele.p, d_b_oth.p, h_np);
    np_pending = true;                 // the count is read after post_neighbor's closing sync()
    HIP_TRY(hipGetLastError());
  }

  // fix_conp.cpp:468-539 post_neighbor
  void post_neighbor(const conp_atoms *at) {
    if (!idx.initialised) throw ConpError(CONP_ERR_STATE, "post_neighbor before setup_post_neighbor");
    if (!have_blist) throw ConpError(CONP_ERR_STATE, "post_neighbor: no neighbor list (init_list not called)");
    resident_step = -1;
    pp_elyte_valid = pp_u_valid = false;
    static const bool tren = getenv("CONP_TIME_REN") != nullptr;
    auto tm0 = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
      if (!tren) return;
      const auto t = std::chrono::steady_clock::now();
      std::fprintf(stderr, "  post_neighbor %-22s %8.1f us\n", what, std::chrono::duration<double, std::micro>(t - tm0).count());
      tm0 = t;
    };
    bool elyte_grew = false;
    // The flattened half list goes to the device as it is: the post-force kernel walks it (one wavefront per owner), and the
    // electrode rows of the real-space b are regrouped from it on the device (conp_rows.hip: count, scan, emit, stable sort).
    // It is the bulk of what a re-neighbour sends (1.4 of 2.7 MB at the headline size, ~0.1 ms of PCIe time): its transfer is
    // started FIRST and runs under the host's bookkeeping below.
    {
      size_t nneigh = 0;
      int iown = 0;                       // list owners are owned atoms: the per-atom tables are needed up to the largest owner only
      for (int ii = 0; ii < blist.inum; ++ii) {
        const int i = blist.ilist[ii];
        nneigh = std::max(nneigh, (size_t)blist.first[i] + (size_t)blist.numneigh[i]);
        iown = std::max(iown, i + 1);
      }
      bl_inum = blist.inum;
      bl_nneigh = nneigh;
      // everything this re-neighbour uploads (bytes); an arena that turns out too small only means that the rest goes the
      // blocking way
      const size_t na = (size_t)at->nlocal + at->nghost;
      ren_begin(sizeof(int) * (4 * na + 6 * (size_t)std::max(idx.elenum_all, 64) + (size_t)blist.inum + 2 * (size_t)iown + nneigh + 4096) + 64 * 32);
      const bool locked = host_page_locked(blist.neigh) && host_page_locked(blist.ilist) && host_page_locked(blist.numneigh) &&
                          host_page_locked(blist.first);
      ren_upload(d_bl_neigh, blist.neigh, std::max<size_t>(nneigh, 1), locked);
      ren_upload(d_bl_ilist, blist.ilist, (size_t)blist.inum, locked);
      ren_upload(d_bl_numneigh, blist.numneigh, (size_t)std::max(iown, 1), locked);
      ren_upload(d_bl_first, blist.first, (size_t)std::max(iown, 1), locked);
    }
    mark("list upload");
    const bool grew = idx.post_neighbor(at->nlocal, at->tag, at->echeck, &elyte_grew, &rc);
    const int ne = idx.elenum_all;
    if (grew) {
      ne_pad = (ne + 127) / 128 * 128;
      tables_current = false;          // the electrode tables are [..][ne_pad] of the OLD electrode count: whoever needs them next rebuilds them
      d_A.reserve((size_t)ne * ne);
      // electrode rows in blocks of ceil(Ne / nranks): rank r's rows start at r * rows_per, so that an all-gather of rows_per
      // values per rank lands every row at its own index (the buffers hold nranks * rows_per >= Ne entries)
      rows_per = (ne + env.nranks - 1) / env.nranks;
      const size_t nvec = std::max<size_t>(ne_pad, (size_t)rows_per * env.nranks);
      d_bk.reserve(4 * (size_t)ne_pad); d_breal.reserve(ne_pad); d_b_own.reserve(nvec); d_eleallq_own.reserve(nvec); d_qele.reserve(ne_pad);
      d_elesetq.reserve(ne_pad); d_eleinitq.reserve(ne_pad); d_ele_z.reserve(ne_pad); d_elecheck.reserve(ne_pad);
      d_ainve.reserve(ne_pad);
      d_bk.zero(stream); d_breal.zero(stream); d_b_own.zero(stream); d_eleallq_own.zero(stream); d_qele.zero(stream);
      d_elesetq.zero(stream); d_eleinitq.zero(stream);
      // own vectors may just have been re-allocated (more electrode atoms than at the last re-neighbouring): a pointer kept from
      // before would dangle.  Only vectors the host bound itself (conp_fix_bind_device_buffers) are left alone.
      if (!b_bound) d_b = d_b_own.p;
      if (!q_bound) d_eleallq = d_eleallq_own.p;
      drop_graph();
      row0 = std::min(ne, env.rank * rows_per);
      row1 = std::min(ne, row0 + rows_per);
    }
    mark("index maps");
    upload_atoms_static(at);
    mark("atoms static");
    map_ghosts(at);
    mark("ghost map");
    build_elyte_list(at, true);
    mark("electrolyte list + schedule");
    build_b_rows_device(at);
    mark("b rows (device)");
    nlocal_cur = at->nlocal;
    sync();
    if (np_pending) { n_b_pairs = *h_np; np_pending = false; }
    mark("sync");
  }

  // Electrolyte atoms that enter the structure factors: `electrode_check == 0 && q != 0` (km_ewald.cpp:685-686).  The reference
  // evaluates that test every step; here the compact list lives on the device between re-neighbourings, and the host-buffer hooks
  // (which see atom->q) re-check the membership at every update -- elyte_list_stale() -- and rebuild the list when an atom's
  // charge has switched between zero and non-zero (fix atom/swap, charge-transfer fixes).  Device-resident hosts
  // (conp_fix_pre_force_device) announce such a change with conp_fix_post_neighbor.
  // ---- the z-window form of the structure-factor contraction (conp_zn.hip; round 5) -----------------------------------------------
  // Used when the handle projects on z classes (planar electrodes), is not spatially decomposed and the list is long enough for 16
  // consecutive atoms of the z-ordered list to share a window of a few grid points (zn_min_atoms).  The electrolyte list is then kept
  // in the order of the atoms' z cells (a stable counting sort at every list build: positions as they are at the re-neighbouring);
  // between re-neighbourings atoms drift by less than the neighbour skin, which the window margins absorb; a tap outside its window
  // raises a device flag that the next synchronisation point reads (the update is repeated on the classic path, the path stays off
  // until the next list build).
  static constexpr int ZN_W = 15;
  static constexpr double ZN_DRIFT = 2.5;       // Angstrom of z drift between list builds the margins allow for
  int zn_min_atoms = 8192;
  bool zn_listed = false;                        // the device list is z-ordered and the items below exist
  bool zn_off = false;                           // a window overflowed since the last list build
  bool zn_tables_current = false;
  int zn_n = 0, zn_ncf = 2;
  double zn_gscale = 0.0, zn_beta = 0.0;
  std::vector<ZnItem> zn_items_h;
  DevBuf<ZnItem> d_zn_items;
  DevBuf<int> d_zn_g0c, d_zn_frag_ptr;
  DevBuf<int2> d_zn_frag_ents;
  DevBuf<double> d_zn_P, d_zn_phihat, d_zn_Bt, d_zn_pieces, d_zn_grid, d_zn_Dt;
  DevBuf<int> d_zn_cov_ptr;
  DevBuf<int2> d_zn_cov_ent;
  int zn_nrg = 0;                 // ranges of the current item list
  DevBuf<double2> d_zn_cs;
  int zn_nfrag = 0;
  std::vector<int> zn_ch_lo, zn_ch_hi;      // per chunk of the ordered list: lowest / highest first tap relative to zn_c_start
  int zn_c_start = 0;
  int *zn_flag_host = nullptr, *zn_flag_dev = nullptr;      // a page-locked word the window kernel stores 1 into when a tap leaves its window
  bool zn_eligible() const { return !decomposed && !args.pppm && nl >= zn_min_atoms && !path_on(CONP_PATH_SK_CLASSIC); }
  bool zn_use() const { return zn_listed && !zn_off && zn_eligible() && sk_projects() && nzc > 0; }
  // rough electrodes (no z classes; one rank): the same contraction, the ranges' raw windows -> rows on the z grid -> G (conp_zn.hip)
  bool zn_gen_use() const { return zn_listed && !zn_off && zn_eligible() && !sk_projects() && nzc == 0 && env.nranks <= 1 && zn_nrg > 0; }
  double zn_lz() const { return kt.slabflag ? env.zprd * env.slab_volfactor : env.zprd; }
  // z-order the list (stable: atoms of one cell keep their list order) and cut it into ranges with a window origin each
  void zn_order_list(const conp_atoms *at) {
    zn_listed = false; zn_off = false;
    if (getenv("CONP_TIME_REN")) std::fprintf(stderr, "  z-window: eligible %d (decomposed %d pppm %d nl %d) plan.nz %d\n", (int)zn_eligible(), (int)decomposed, args.pppm, nl, plan.nz);
    if (!zn_eligible() || plan.nz <= 0) return;
    // grid: the smallest multiple of 16 with an oversampling ratio n / (2 nz) >= 1.9 (15 taps: 2e-13 of the largest entry there,
    // tools/proto/zn_proto.py); the window's shape parameter follows the ratio
    int n = std::max(64, (38 * plan.nz / 10 + 15) / 16 * 16);       // any integer does (no FFT anywhere): a multiple of 16
    zn_n = n;
    zn_beta = 0.97 * 3.14159265358979323846 * ZN_W * (1.0 - (double)plan.nz / n);      // gamma pi W (1 - 1 / (2 sigma))
    const double lz = zn_lz();
    zn_gscale = (double)n / lz;
    const int nlist = (int)elyte_idx_h.size();
    std::vector<int> cell(nlist), occ(n, 0), cnt(n + 1, 0);
    std::vector<double> uw(nlist);                       // grid coordinate of every listed atom, wrapped into [0, n)
    const double rn = 1.0 / n;
    for (int k = 0; k < nlist; ++k) {
      double u = at->x[3 * (size_t)elyte_idx_h[k] + 2] * zn_gscale;
      u -= n * std::floor(u * rn);
      int c = (int)u;                                    // (no integer division in these loops: 32768 atoms x 3 of them were 0.2 ms)
      if (c >= n) { c = n - 1; u = std::nextafter((double)n, 0.0); }
      uw[k] = u; cell[k] = c; ++occ[c];
    }
    // the list starts behind the longest run of empty cells (the vacuum / the electrodes of a slab cell), so that no chunk of 16
    // consecutive atoms straddles it; a box without a gap starts at cell 0 and the windows wrap (positions are taken relative to
    // a range's origin, modulo the grid)
    int best_len = 0, best_end = 0, run = 0;
    for (int c = 0; c < 2 * n; ++c) {
      if (occ[c % n] == 0) { if (++run > best_len && run <= n) { best_len = run; best_end = c % n; } }
      else run = 0;
    }
    const int c_start = best_len > 0 ? (best_end + 1) % n : 0;
    for (int k = 0; k < nlist; ++k) { int c = cell[k] - c_start; if (c < 0) c += n; cell[k] = c; ++cnt[c + 1]; }
    for (int c = 0; c < n; ++c) cnt[c + 1] += cnt[c];
    elyte_dev_h.assign(nlist, 0);
    // per chunk of 16 atoms of the ordered list: the lowest and the highest first tap (grid index relative to the sort origin c_start;
    // the ranges are cut from these, whatever their length -- one pass over the atoms per list build)
    zn_c_start = c_start;
    const int nch = (nlist + 15) / 16;
    zn_ch_lo.assign(std::max(nch, 1), 0x3fffffff); zn_ch_hi.assign(std::max(nch, 1), -0x3fffffff);
    for (int k = 0; k < nlist; ++k) {
      const int pos = cnt[cell[k]]++;
      elyte_dev_h[pos] = elyte_idx_h[k];
      double ur = uw[k] - c_start;
      if (ur < 0.0) ur += n;                                              // [0, n): the ordered list does not wrap
      const int i0 = (int)std::ceil(ur - 0.5 * ZN_W);
      int &lo = zn_ch_lo[pos >> 4], &hi = zn_ch_hi[pos >> 4];
      lo = std::min(lo, i0); hi = std::max(hi, i0);
    }
    // ranges of chunks: about three workgroups of four waves per CU over all row tiles; the rank's share of the chunk axis comes
    // from build_items (table_c0 .. table_c1)
    zn_listed = true;
  }
  void zn_build_items(const conp_atoms *at) {
    if (!zn_listed) return;
    const int n = zn_n;
    const int nchunks = nl_pad / 16;
    const int c_lo = env.nranks > 1 ? table_c0 : 0, c_hi = env.nranks > 1 ? table_c1 : nchunks;
    const int nrt = std::max(1, (int)own_rt_h.size());
    const double cellw = zn_lz() / n;
    const int margin = (int)std::ceil(ZN_DRIFT / cellw);
    std::vector<int> g0c(std::max(nchunks, 1), 0);
    zn_items_h.clear();
    int need_max = 0;
    std::vector<std::pair<int, int>> ranges;
    // All workgroups of the launch are resident at once (four per CU with 32 window columns, three with 48: LDS) and equally long,
    // so their number should FILL the slots: nr * nrt just below slots (86 ranges x 9 row tiles = 774 on 768 slots left six CUs with
    // a fourth workgroup and the launch a third longer).  Large boxes take more ranges (a range should not span more than ~8 cells).
    // candidates in order of preference: 32 columns with one full launch, 48 columns with one, then two and three rounds of each
    // (shorter ranges span fewer cells; a range keeps at least six chunks)
    const int cand[][2] = {{4, 1}, {4, 2}, {3, 1}, {4, 3}, {4, 4}, {3, 2}, {4, 6}, {4, 8}, {3, 3}, {3, 4}, {3, 6}, {3, 8}};
    for (const auto &cd : cand) {
      const int per_cu = cd[0];
      int nr = std::max(1, per_cu * num_cus / nrt) * cd[1];
      nr = std::min(nr, std::max(1, (c_hi - c_lo) / 6));
      ranges.clear(); need_max = 0;
      for (int r = 0; r < nr; ++r) {
        const int a = c_lo + (int)((long long)(c_hi - c_lo) * r / nr), b = c_lo + (int)((long long)(c_hi - c_lo) * (r + 1) / nr);
        if (b <= a) continue;
        int imin = 0x3fffffff, imax = -0x3fffffff;
        for (int c = a; c < b && c < (int)zn_ch_lo.size(); ++c) { imin = std::min(imin, zn_ch_lo[c]); imax = std::max(imax, zn_ch_hi[c]); }
        if (imin > imax) { imin = imax = 0; }                 // (a range of padding atoms only)
        const int g0 = zn_c_start + imin - margin;
        need_max = std::max(need_max, imax - imin + ZN_W + 2 * margin);
        for (int c = a; c < b; ++c) g0c[c] = g0;
        ranges.push_back({a, b});
      }
      if (need_max <= (per_cu == 4 ? 32 : 48)) break;
    }
    if (getenv("CONP_TIME_REN")) std::fprintf(stderr, "  z-window: n %d, %zu ranges x %d row tiles, need %d columns, margin %d\n", n, ranges.size(), nrt, need_max, margin);
    if (need_max > 48 || ranges.empty()) { zn_listed = false; return; }     // too sparse for a window: the classic kernels
    zn_ncf = need_max <= 32 ? 2 : 3;
    // Launch order: workgroups go to the eight XCDs round-robin (block b -> XCD b mod 8, each with an L2 of its own), and the row
    // tiles of one range read the same phase tables and window matrix -- so all row tiles of range r run on XCD r mod 8: an XCD pulls
    // an eighth of the tables through the fabric once and finds them in its L2 for the other row tiles.  The pieces' slots keep
    // the (row tile, range) order the sums are formed in.
    const int nrg = (int)ranges.size();
    std::vector<ZnItem> by_slot;
    int slot = 0;
    for (int k = 0; k < nrt; ++k)
      for (auto &rg : ranges) {
        const int rt = own_rt_h.empty() ? 0 : own_rt_h[k];
        by_slot.push_back(ZnItem{rt, rg.first, rg.second, g0c[rg.first], slot++, rt >= plan.paired_lo && rt < plan.paired_hi ? 1 : 0});
      }
    {
      std::vector<std::vector<int>> per_xcd(8);
      for (int r = 0; r < nrg; ++r)
        for (int k = 0; k < nrt; ++k) per_xcd[r & 7].push_back(k * nrg + r);
      size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      zn_items_h.reserve(by_slot.size());
      while (zn_items_h.size() < by_slot.size())
        for (int x = 0; x < 8; ++x)
          if (pos[x] < per_xcd[x].size()) zn_items_h.push_back(by_slot[per_xcd[x][pos[x]++]]);
          else {
            // this XCD has run out of items of its own: it takes one from the XCD with the most left (the launch must stay dense)
            int best = -1; size_t left = 0;
            for (int y = 0; y < 8; ++y) if (per_xcd[y].size() - pos[y] > left) { left = per_xcd[y].size() - pos[y]; best = y; }
            if (best >= 0) zn_items_h.push_back(by_slot[per_xcd[best][pos[best]++]]);
          }
    }
    // the pieces of a row fragment, range after range: hc_sum_kernel's lists (a row tile's piece = a band of four row fragments)
    zn_nfrag = 4 * plan.n_row_tiles;
    std::vector<std::vector<int2>> of_frag(zn_nfrag);
    for (const ZnItem &it : zn_items_h)
      for (int f = 0; f < 4; ++f) of_frag[4 * it.rt + f].push_back(make_int2(it.slot * sk_hc_stride() + 16 * f, 4));
    std::vector<int> fptr(zn_nfrag + 1, 0);
    std::vector<int2> fent;
    for (int g = 0; g < zn_nfrag; ++g) { fent.insert(fent.end(), of_frag[g].begin(), of_frag[g].end()); fptr[g + 1] = (int)fent.size(); }
    if (fent.empty()) fent.push_back(make_int2(0, 4));
    // rough electrodes: which (range, column) pairs hold grid point g, range after range
    zn_nrg = nrg;
    if (nzc == 0) {
      const int ncol = 16 * zn_ncf;
      std::vector<int> cptr(n + 1, 0);
      std::vector<int2> cent;
      for (int g = 0; g < n; ++g) {
        for (int r = 0; r < nrg; ++r) {
          int col = (g - g0c[ranges[r].first]) % n;
          if (col < 0) col += n;
          if (col < ncol) cent.push_back(make_int2(r, col));
        }
        cptr[g + 1] = (int)cent.size();
      }
      if (cent.empty()) cent.push_back(make_int2(0, 0));
      ren_upload(d_zn_cov_ptr, cptr);
      ren_upload(d_zn_cov_ent, cent);
    }
    ren_upload(d_zn_items, zn_items_h);
    ren_upload(d_zn_g0c, g0c);
    ren_upload(d_zn_frag_ptr, fptr);
    ren_upload(d_zn_frag_ents, fent);
    d_zn_Bt.reserve((size_t)nchunks * 48 * 16);
    d_zn_pieces.reserve(std::max<size_t>(1, zn_items_h.size()) * std::max<size_t>(sk_hc_stride(), nzc == 0 ? (size_t)16 * zn_ncf * 128 : 0));
    if (nzc == 0) d_zn_grid.reserve((size_t)nrt * n * 128);
    if (!zn_flag_host) {
      HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&zn_flag_host), 64, hipHostMallocMapped));
      HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&zn_flag_dev), zn_flag_host, 0));
    }
    *zn_flag_host = 0;
  }
  // a window overflowed in an update that has completed: the classic kernels from now on (until the next list build); true = the
  // caller has results of this handle in flight or in hand that may be wrong
  bool zn_overflowed() {
    if (!zn_flag_host || *zn_flag_host == 0) return false;
    *zn_flag_host = 0;
    zn_off = true;
    mesgf("conp/hip: an electrolyte atom left the z-window of its place in the list (more than %.1f A of z drift since the last "
          "re-neighbouring); the update is repeated with the full structure-factor kernels\n", ZN_DRIFT);
    return true;
  }
  // once per plan / z-class table: the window's Fourier transform (Gauss-Legendre quadrature on the host), the grid's phases, P
  void zn_ensure_tables() {
    if (zn_tables_current) return;
    const int n = zn_n, nzm = plan.nz;
    const double h = 6.283185307179586476925286766559 / n, a = 0.5 * ZN_W * h, beta = zn_beta;
    // nodes and weights of the 64-point Gauss-Legendre rule by Newton iteration on P_64
    const int Q = 64;
    std::vector<double> xs(Q), ws(Q);
    for (int i = 0; i < Q; ++i) {
      double x = std::cos(3.14159265358979323846 * (i + 0.75) / (Q + 0.5)), dp = 0.0;
      for (int itn = 0; itn < 100; ++itn) {
        double p0 = 1.0, p1 = x;
        for (int k = 2; k <= Q; ++k) { const double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k; p0 = p1; p1 = p2; }
        dp = Q * (x * p1 - p0) / (x * x - 1.0);
        const double dx = p1 / dp;
        x -= dx;
        if (std::fabs(dx) < 1e-16) break;
      }
      xs[i] = x; ws[i] = 2.0 / ((1.0 - x * x) * dp * dp);
    }
    std::vector<double> phihat(nzm);
    for (int m = 0; m < nzm; ++m) {
      double sum = 0.0;
      for (int i = 0; i < Q; ++i) sum += ws[i] * a * std::exp(beta * (std::sqrt(1.0 - xs[i] * xs[i]) - 1.0)) * std::cos(m * a * xs[i]);
      phihat[m] = sum;
    }
    std::vector<double2> cs(n);
    for (int k = 0; k < n; ++k) cs[k] = make_double2(std::cos(h * k), std::sin(h * k));
    d_zn_phihat.upload(phihat, stream); d_zn_cs.upload(cs, stream);
    if (nzc > 0) {
      d_zn_P.reserve((size_t)plan.R_pad / 2 * nzc * n);      // one row per planar vector
      launch_zn_ptable(stream, dplan, plan.kzt, nzc, n, d_TzcT.p, d_zn_phihat.p, d_zn_cs.p, d_zn_P.p);
    } else {
      d_zn_Dt.reserve((size_t)n * plan.C_pad);
      launch_zn_dtable(stream, dplan, plan.kzt, n, d_zn_phihat.p, d_zn_cs.p, d_zn_Dt.p);
    }
    sync();                                          // (the host vectors go out of scope)
    zn_tables_current = true;
  }
  std::vector<int> elyte_dev_h;                    // the list as the device holds it (z-ordered when zn_listed)
  void build_elyte_list(const conp_atoms *at, bool inside_reneighbour = false) {
    if (!inside_reneighbour) ren_off = 0;      // (the callers outside a re-neighbour have synchronised the stream: the arena is free)
    elyte_idx_h.clear();
    for (int i = 0; i < at->nlocal; ++i) if (at->echeck[i] == 0 && at->q[i] != 0) elyte_idx_h.push_back(i);
    nl_local = nl = (int)elyte_idx_h.size();
    if (decomposed) {
      // every rank's charged electrolyte atoms enter every rank's structure factors: their (x, q) are all-gathered at each
      // update (b_cal) into d_xg / d_qg, rank-major; the phase kernel then walks that compact array
      elyte_counts.assign(env.nranks, 0);
      rc.allgather_int(nl_local, elyte_counts.data());
      nl = 0;
      for (int v : elyte_counts) nl += v;
      std::vector<int> iota(std::max(nl, 1));
      for (int i = 0; i < nl; ++i) iota[i] = i;
      d_iota.upload(iota, stream);
      d_xg.reserve((size_t)std::max(nl, 1) * 3); d_qg.reserve(std::max(nl, 1));
    }
    // atoms are consumed in chunks of 32; the splits want an even share of chunks
    nl_pad = std::max(32, (nl + 31) / 32 * 32);
    static const bool tren = getenv("CONP_TIME_REN") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
      if (!tren) return;
      const auto t1 = std::chrono::steady_clock::now();
      std::fprintf(stderr, "    electrolyte list: %-18s %7.1f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count());
      t0 = t1;
    };
    lap("scan");
    zn_order_list(at);
    lap("z order");
    ren_upload(d_elyte_idx, zn_listed ? elyte_dev_h : elyte_idx_h);
    build_items();
    lap("sk_gemm schedule");
    zn_build_items(at);
    lap("z-window items");
    d_Xt.reserve((size_t)(plan.kxmax + 2) * nl_pad); d_Yt.reserve((size_t)(plan.kymax + 1) * nl_pad);
    d_Zt.reserve((size_t)(1 + plan.n_col_tiles * 32) * nl_pad); d_Zt.zero(stream);   // unit step + a seed every 5th kz
    d_qc.reserve(nl_pad); d_slab_part.reserve((nl_pad + 31) / 32 + 1025);      // (one per z block of the phase kernel: up to four per 128 atoms)
    // partial tiles: fragments beyond a tile's sphere cut are never written -- all zero from the start, finite ever after
    reserve_partials();
  }
  // After an update whose sk_gemm projected its tiles (planar electrodes) G was never formed: the phase tables of that update are
  // still on the device, so the contraction is run again in the partial-tile mode and reduced (structure factors are a diagnostic
  // getter: conp_fix_get_sfac, KSpaceModule::sfac in the reference's energy output).
  // (the z-window form writes no z phase seeds: the full kernels' tables of the last update are made first -- same atoms, same list)
  bool zs_current = true;
  const double *last_ex = nullptr, *last_eq = nullptr;
  const int *last_eidx = nullptr;
  void refresh_structure_factors() {
    if (g_current) return;
    if (!zs_current && last_ex) {
      int nsp = 0;
      launch_elyte_phase(stream, nl, nl_pad, last_eidx, last_ex, last_eq, kt.unitk[0], kt.unitk[1], kt.unitk[2], plan.kxmax, plan.kymax, plan.nz,
                         plan.kzt, 1 + plan.n_col_tiles * 32, d_Xt.p, d_Yt.p, d_Zt.p, d_qc.p, d_slab_part.p, &nsp, nullptr, d_breal.p,
                         16 * table_c0, 16 * table_c1);
      zs_current = true;
    }
    reserve_partials(true);
    launch_sk_gemm(stream, dplan, d_witems.p, witems_maxseg, (int)seg_ptr_h.size() - 1, nl_pad, d_Xt.p, d_Yt.p, d_Zt.p, d_qc.p,
                   d_Gpart.p);
    launch_sk_reduce(stream, dplan, d_tiles.p, (int)tiles_h.size(), max_nsplit, d_Gpart.p, d_G.p, d_Gw.p);
    g_current = true;
  }
  // sk_gemm's output per segment: the projected piece (planar electrodes) or the partial tile.  The partial tiles of a projecting
  // handle are allocated when somebody asks for the structure factors (conp_fix_get_sfac).
  bool sk_projects() const {
    return nzc > 0 && nzc <= sk_hc_max_classes() && !args.pppm && !sk_partials && !no_fuse;
  }
  void reserve_partials(bool tiles_too = false) {
    const size_t ns = std::max<size_t>(1, (size_t)n_slots);
    if (sk_projects()) d_Hpart.reserve(std::max<size_t>(1, items_h.size()) * sk_hc_stride());
    if (!sk_projects() || tiles_too) {
      // partial tiles: fragments beyond a tile's sphere cut are never written -- all zero from the start, finite ever after.  A
      // projecting handle cuts bands that straddle row tiles: a slot then holds the fragments of ONE band, the others must read as
      // zero -- it asks for partial tiles only for conp_fix_get_sfac, and clears them every time.
      const bool grew = ns * 128 * 320 > d_Gpart.n;
      if (grew) d_Gpart.reserve(ns * 128 * 320);
      if (grew || sk_projects()) d_Gpart.zero(stream);
    }
  }
  bool elyte_list_stale(const conp_atoms *at) {
    size_t k = 0;
    const size_t n = elyte_idx_h.size();
    int stale = 0;
    for (int i = 0; i < at->nlocal && !stale; ++i) {
      if (at->echeck[i] != 0 || at->q[i] == 0) continue;
      if (k >= n || elyte_idx_h[k] != i) stale = 1;
      ++k;
    }
    if (!stale && k != n) stale = 1;
    rc.allreduce_max_int(&stale, 1);       // decomposed: the gather layout is common to all ranks, so is the decision
    return stale != 0;
  }
  // decomposed runs: this rank's charged electrolyte atoms -> everybody (MPI_Allgatherv), then onto the device
  void gather_elyte(const conp_atoms *at) {
    xq_pack.resize((size_t)nl_local * 4);
    for (int k = 0; k < nl_local; ++k) {
      const int i = elyte_idx_h[k];
      xq_pack[4 * (size_t)k] = at->x[3 * (size_t)i]; xq_pack[4 * (size_t)k + 1] = at->x[3 * (size_t)i + 1];
      xq_pack[4 * (size_t)k + 2] = at->x[3 * (size_t)i + 2]; xq_pack[4 * (size_t)k + 3] = at->q[i];
    }
    xq_all.resize((size_t)std::max(nl, 1) * 4);
    rc.gatherv(xq_pack.data(), elyte_counts, 4, xq_all.data());
    // de-interleave into the layouts the kernels read: x [nl][3], q [nl]
    double *st = pinned((size_t)ne_pad + 8 + (size_t)nl * 4) + ne_pad + 8;
    sync();
    for (int k = 0; k < nl; ++k) {
      st[3 * (size_t)k] = xq_all[4 * (size_t)k]; st[3 * (size_t)k + 1] = xq_all[4 * (size_t)k + 1];
      st[3 * (size_t)k + 2] = xq_all[4 * (size_t)k + 2]; st[3 * (size_t)nl + k] = xq_all[4 * (size_t)k + 3];
    }
    if (nl > 0) {
      HIP_TRY(hipMemcpyAsync(d_xg.p, st, (size_t)nl * 3 * sizeof(double), hipMemcpyHostToDevice, stream));
      HIP_TRY(hipMemcpyAsync(d_qg.p, st + 3 * (size_t)nl, (size_t)nl * sizeof(double), hipMemcpyHostToDevice, stream));
    }
  }

  // FixConp::b_comm (fix_conp.cpp:641-648) for w values per atom: rows of the owned electrode atoms, in `ele` order, from every
  // rank (MPI_Allgatherv with elenum_list / displs) land at their permanent index through elebuf2eleall
  std::vector<double> ele_comm_buf;
  void ele_comm(const double *send /*[elenum][w]*/, int w, double *recv /*[Ne][w]*/) {
    const int ne = idx.elenum_all;
    ele_comm_buf.resize((size_t)std::max(ne, 1) * w);
    rc.gatherv(send, idx.elenum_list, w, ele_comm_buf.data());
    for (int iall = 0; iall < ne; ++iall)
      for (int c = 0; c < w; ++c) recv[(size_t)idx.elebuf2eleall[iall] * w + c] = ele_comm_buf[(size_t)iall * w + c];
  }

  // sk_gemm schedule ("stream-K" over the atom chunks): the work of all tiles of this rank is laid out on one axis,
  // tile after tile, chunk after chunk, with cost (mean nba of the 4 row fragments + SK_C0) per chunk of 16 atoms (MFMA work ~ nba, operand
  // generation + barrier ~ SK_C0), and cut into num_cus equal shares.  A share is a list of segments (tile, chunk
  // range); every segment writes one partial tile, sk_reduce adds a tile's segments in order.
  // The two constants are a least-squares fit of per-segment lengths measured on the headline box (tools/sk_stamp.py ->
  // tools/sk_fit.py; round 3's kernel:  us = 0.483 * chunks * (mean kz blocks + 1.37) + 2.8 per segment): a segment's start (first
  // panel with nothing to overlap it) and its partial-tile write cost as much as 5.74 units of chunk cost.  The fit leaves 1 % rms
  // per segment; what remains between workgroups (max / median 1.05 on the box measured) is the speed of the XCD a workgroup
  // lands on (+-3 %, different XCDs on different boxes) -- nothing a static plan can see, and an adaptive one would give up the
  // run-to-run reproducibility of the bits.  Round 1's model (C0 = 2, no
  // per-segment term) left the heavy tiles' workgroups and those whose share straddles a tile boundary 3 % behind the rest.
  // (Tried on top: a linear ramp of the shares so that early finishers' partial-tile stores overlap the others' last chunks --
  //  no effect at +-8 / 16 / 24 units, 244.1 - 244.5 us.  What is left is a +-2 % spread between XCDs.)
  static double diag_number(const char *v, double dflt) { return v ? atof(v) : dflt; }      // (diag_switch folds to null in the product)
  double SK_C0 = diag_number(diag_switch("CONP_SK_C0"), 1.37);
  double SK_CSEG = diag_number(diag_switch("CONP_SK_CSEG"), 5.74);
  // what the schedule was cut for: the same padded atom count, plan and output form give the same schedule -- a re-neighbour that
  // changes none of them (the usual one) keeps the work list that is on the device
  struct ItemsKey { int nl_pad = -1; long plan_gen = -1; bool proj = false; int nzc = -1; int nranks = 0; bool operator==(const ItemsKey &o) const {
    return nl_pad == o.nl_pad && plan_gen == o.plan_gen && proj == o.proj && nzc == o.nzc && nranks == o.nranks; } };
  ItemsKey items_key;
  long plan_gen = 0;
  void build_items() {
    {
      const ItemsKey k{nl_pad, plan_gen, sk_projects(), nzc, env.nranks};
      if (k == items_key && !items_h.empty()) return;
      items_key = k;
    }
    const int nchunks = nl_pad / 16;
    // this rank's chunk range of every tile (all of it on one rank; km_conp_setup); a tile may come out empty
    const size_t ntiles = tiles_h.size();
    std::vector<int> tclo(ntiles, 0), tchi(ntiles, nchunks);
    bool uniform = true;                   // every tile with the same chunk range (one rank: all of it; N ranks: the rank's atoms)
    for (size_t i = 0; i < ntiles; ++i) {
      tclo[i] = (int)std::lround(tile_flo[i] * nchunks);
      tchi[i] = (int)std::lround(tile_fhi[i] * nchunks);
      uniform = uniform && tclo[i] == tclo[0] && tchi[i] == tchi[0];
    }
    // BANDS (round 4).  sk_gemm's unit of work is a band of planar vectors x a column tile: rf consecutive row fragments (16 planar
    // vectors each) of the plan.  A wave holds 20 accumulator fragments either as 4 row x 5 column fragments -- a row tile of the
    // plan, what every band was until round 3 -- or as 5 x 4: a column tile that uses at most 16 of its 20 column fragments (the
    // headline box: nz = 126; the slab geometry: 15 per tile) is cut into bands of FIVE row fragments, 80 planar vectors instead of
    // 64 per panel and barrier: 33 row fragments = 5 x 5 + 2 x 4 -> 7 bands instead of 9 tiles (the ninth a runt of 9 vectors that
    // paid a tile's full per-chunk cost), and 80 instead of 64 MFMAs per wave and chunk where the sphere is full.  A band that
    // straddles two row tiles of the plan writes one output slot per tile (zero rows where the other band writes): everything
    // behind sk_gemm still sees row tiles.  Only the projecting mode takes bands of five (its slots are 8-KB pieces; partial tiles
    // would double sk_reduce's reads); CONP_SK_BANDS4: comparison switch, bands = row tiles everywhere.
    struct Band { int g0, rf, ct; unsigned long long nbf; int clo, chi; };
    std::vector<Band> bands;
    {
      std::map<std::pair<int, int>, size_t> tile_at;
      for (size_t i = 0; i < ntiles; ++i) tile_at[{tiles_h[i].rt, tiles_h[i].ct}] = i;
      const bool five = sk_projects() && uniform && ntiles > 0 && diag_switch("CONP_SK_BANDS4") == nullptr;
      for (int ct = 0; ct < plan.n_col_tiles; ++ct) {
        const int nfr = 4 * plan.n_row_tiles;
        std::vector<int> nfa(nfr + 8, 0);
        int n = 0;
        for (int g = 0; g < nfr; ++g) {
          auto it = tile_at.find({g >> 2, ct});
          if (it == tile_at.end()) continue;
          nfa[g] = (int)((tiles_h[it->second].nbf >> (8 * (g & 3))) & 255u);
          if (nfa[g] > 0) n = g + 1;
        }
        if (n == 0) continue;
        if (!five) {
          for (int rt = 0; 4 * rt < n; ++rt) {
            auto it = tile_at.find({rt, ct});
            if (it == tile_at.end()) continue;
            bands.push_back(Band{4 * rt, 4, ct, (unsigned long long)tiles_h[it->second].nbf, tclo[it->second], tchi[it->second]});
          }
          continue;
        }
        // fewest bands covering the fragments [0, n) with pieces of 4 or 5 (5 only where no fragment has more than 16 column
        // fragments), then the least padding behind n; ties: fives first (the early, full fragments)
        std::vector<int> best(n + 6, 1 << 20), pad(n + 6, 1 << 20), take(n + 6, 0);
        for (int i = n + 5; i >= 0; --i) {
          if (i >= n) { best[i] = 0; pad[i] = i - n; continue; }
          for (int rf : {5, 4}) {
            bool ok = true;
            if (rf == 5) for (int f = 0; f < 5; ++f) ok = ok && nfa[i + f] <= 16;
            if (!ok) continue;
            const int j = std::min(i + rf, n + 5);
            if (best[j] + 1 < best[i] || (best[j] + 1 == best[i] && pad[j] < pad[i])) { best[i] = best[j] + 1; pad[i] = pad[j]; take[i] = rf; }
          }
        }
        const size_t t0 = tile_at.begin()->second;
        for (int i = 0; i < n; i += take[i]) {
          unsigned long long nbf = 0;
          for (int f = 0; f < take[i]; ++f) nbf |= (unsigned long long)(i + f < nfr ? nfa[i + f] : 0) << (8 * f);
          bands.push_back(Band{i, take[i], ct, nbf, tclo[t0], tchi[t0]});
        }
      }
    }
    const size_t nt = bands.size();
    // a segment that ends in the projecting epilogue costs about twice one that stores its partial tile (stamped build: 15 units;
    // A/B on one rank: 5.74 / 9 / 12 equal within noise, 15 worse; on emulated ranks with 16 chunks per workgroup 12-14 is 7 % faster)
    if (!diag_switch("CONP_SK_CSEG")) SK_CSEG = sk_projects() ? 12.0 : 5.74;
    // one workgroup per CU, except for small problems: sk_reduce walks a tile's splits serially (~1 us per split), so a tile
    // is cut into more than 16 segments only when a segment still holds >= 8 chunks (measured on the decks: il_onelayer
    // 57 -> 52 us per update with 32 instead of 256 workgroups)
    // With the projecting epilogue (planar electrodes) a segment leaves 8 KB, not a partial tile, and the dot kernel adds up to 32
    // pieces per row tile itself: twice as many workgroups pay (il_onelayer 30.9 -> 29.4 us per update with 64; 128: 30.0).
    int nwg = std::max(1, num_cus);
    nwg = std::min(nwg, std::max((int)((sk_projects() ? 32 : 16) * nt), (int)((nt * (size_t)nchunks + 7) / 8)));
    nwg = std::max(1, nwg);
    if (debug_sk_workgroups() > 0) nwg = debug_sk_workgroups();
    // MFMA work of a band ~ its row fragments' active column fragments (per-fragment sphere culling), in the unit the constants were
    // fitted in: kz blocks of 16 of a four-fragment tile (two column fragments each)
    auto cost = [&](const Band &t) {
      double sum = 0.0;
      for (int f = 0; f < t.rf; ++f) sum += (double)((t.nbf >> (8 * f)) & 255u);
      return 0.125 * sum + SK_C0;
    };
    struct Seg { int tile, c0, c1, wg; };
    std::vector<Seg> segs;
    // cuts the chunk ranges [clo, chi) of the tiles, laid end to end, into `nshare` shares of equal cost; share s goes to
    // workgroup wg0 + s * wgstride
    auto cut = [&](const std::vector<int> &clo, const std::vector<int> &chi, int nshare, int wg0, int wgstride) {
      // work left from (tile ti, chunk ch) to the end, without segment starts
      std::vector<double> tail(nt + 1, 0.0);
      for (size_t i = nt; i-- > 0;) tail[i] = tail[i + 1] + std::max(0, chi[i] - clo[i]) * cost(bands[i]);
      size_t total = 0;
      for (size_t i = 0; i < nt; ++i) total += (size_t)std::max(0, chi[i] - clo[i]);
      // a segment shorter than this costs more in start-up than it carries -- but only where shares are long: a small system's
      // share IS one or two chunks (dilute: 16 chunks on 16 workgroups), and merging "slivers" there would quadruple one share
      const int MIN_SEG = std::max(1, std::min(4, (int)(total / (size_t)nshare) / 2));
      size_t ti = 0;
      while (ti < nt && chi[ti] <= clo[ti]) ++ti;
      int ch = ti < nt ? clo[ti] : 0;
      for (int w = 0; w < nshare && nt > 0; ++w) {
        if (ti >= nt) continue;
        const bool last = w + 1 == nshare;
        // equal shares of what is left, counting one segment start per remaining workgroup and one per tile boundary ahead
        const double left = tail[ti] - (ch - clo[ti]) * cost(bands[ti]) + SK_CSEG * ((double)(nshare - w) + (double)(nt - 1 - ti));
        double budget = left / (nshare - w) - SK_CSEG;
        bool first = true;
        while (ti < nt) {
          const double c = cost(bands[ti]);
          const int avail = chi[ti] - ch;
          int take;
          if (last) take = avail;
          else {
            if (!first) budget -= SK_CSEG;                     // starting another segment in this share
            take = (int)std::lround(budget / c);
            if (first) take = std::max(take, 1);
            if (!first && take < MIN_SEG) break;               // not worth a new segment: the next workgroup starts this tile
            if (avail - take < MIN_SEG) take = avail;          // do not leave a sliver of the tile behind
            take = std::min(take, avail);
          }
          segs.push_back(Seg{(int)ti, ch, ch + take, wg0 + w * wgstride});
          budget -= take * c;
          ch += take;
          first = false;
          bool inside = true;                                  // stopped inside a tile?
          if (ch >= chi[ti]) {
            ++ti;
            while (ti < nt && chi[ti] <= clo[ti]) ++ti;
            ch = ti < nt ? clo[ti] : 0;
            inside = false;
          }
          if (!last && (budget < c || inside)) break;          // share used up (or stopped inside a tile)
        }
      }
    };
    std::vector<int> clo(nt, 0), chi(nt, nchunks);
    for (size_t i = 0; i < nt; ++i) { clo[i] = bands[i].clo; chi[i] = bands[i].chi; }
    // the atoms whose phase tables this rank reads (elyte_phase fills those only)
    table_c0 = nchunks; table_c1 = 0;
    for (size_t i = 0; i < nt; ++i) if (chi[i] > clo[i]) { table_c0 = std::min(table_c0, clo[i]); table_c1 = std::max(table_c1, chi[i]); }
    if (table_c1 <= table_c0) table_c0 = table_c1 = 0;
    // XCD-aware shares (the full chip on whole tiles): workgroups are dealt to the eight XCDs round-robin (w mod 8), each XCD has
    // its own L2 and the phase tables are blocked by atom chunk.  XCD x takes the x-th eighth of the atoms of EVERY tile, cut into
    // nwg / 8 shares for its workgroups w = x, x + 8, ...: equal work per XCD by construction, and an XCD pulls only its eighth of
    // the tables through the fabric (round 2: one axis over all tiles -- every XCD read ~70 % of the tables: 5.6 table volumes per
    // launch at the memory side, profiles/r03).  The block -> XCD map is not an architectural promise; nothing but the traffic
    // depends on it.
    // (XCD-aware shares put ~1.2 segments on a workgroup instead of ~1.03: worth it while a share is long -- 72 chunks on one rank --,
    //  not when a rank's share leaves 8-16 chunks per workgroup and a segment's fixed cost is a fifth of it: emulated ranks,
    //  tools/rank_emulation.py, N = 4: 80.9 vs 74.6 us, N = 8: 48.3 vs 41.2)
    const int span = nt ? chi[0] - clo[0] : 0;
    const bool long_shares = (size_t)span * nt >= (size_t)32 * nwg;
    const bool xcd_aware = uniform && long_shares && nwg >= 64 && nwg % 8 == 0 && span >= 64 && diag_switch("CONP_SK_FLAT") == nullptr;
    if (xcd_aware) {
      std::vector<int> lo(nt), hi(nt);
      for (int x = 0; x < 8; ++x) {
        for (size_t i = 0; i < nt; ++i) { lo[i] = clo[0] + (int)((long)span * x / 8); hi[i] = clo[0] + (int)((long)span * (x + 1) / 8); }
        cut(lo, hi, nwg / 8, x, 8);
      }
    } else cut(clo, chi, nwg, 0, 1);
    // segments tile-major (a tile's partial tiles are contiguous: the reducing kernels walk item0 .. item0 + nsplit - 1), and every
    // workgroup's list of segment indices
    std::stable_sort(segs.begin(), segs.end(), [](const Seg &a, const Seg &b) { return a.tile != b.tile ? a.tile < b.tile : a.c0 < b.c0; });
    items_h.clear();
    std::vector<std::vector<int>> per_wg(nwg);
    for (const Seg &g : segs) {
      per_wg[g.wg].push_back((int)items_h.size());
      const Band &bd = bands[g.tile];
      items_h.push_back(SkItem{bd.g0, bd.rf, bd.ct, g.c0, g.c1, bd.nbf});
    }
    seg_ptr_h.assign(nwg + 1, 0);
    seg_idx_h.clear();
    for (int w = 0; w < nwg; ++w) {
      seg_ptr_h[w] = (int)seg_idx_h.size();
      seg_idx_h.insert(seg_idx_h.end(), per_wg[w].begin(), per_wg[w].end());
    }
    seg_ptr_h[nwg] = (int)seg_idx_h.size();
    if (seg_idx_h.empty()) seg_idx_h.push_back(0);
    // OUTPUT SLOTS: one per (segment, row tile of the plan the segment's band has fragments in), numbered tile-major so that a
    // tile's slots are item0 .. item0 + nsplit - 1 for the kernels that add them (sk_reduce, hc_sum, the dot kernel)
    std::vector<int> seg_sga(items_h.size(), -1), seg_sgb(items_h.size(), -1);
    {
      std::map<std::pair<int, int>, size_t> tile_at;
      for (size_t i = 0; i < tiles_h.size(); ++i) tile_at[{tiles_h[i].rt, tiles_h[i].ct}] = i;
      std::vector<std::vector<std::pair<int, int>>> of_tile(tiles_h.size());       // (segment, side)
      for (size_t sgm = 0; sgm < items_h.size(); ++sgm) {
        const SkItem &it = items_h[sgm];
        for (int side = 0; side < 2; ++side) {
          const int rt = (it.g0 >> 2) + side;
          bool any = false;                       // does the band hold a real fragment of this row tile?
          for (int f = 0; f < it.rf; ++f) any = any || (((it.g0 + f) >> 2) == rt && ((it.nbf >> (8 * f)) & 255u) != 0);
          auto at_ = tile_at.find({rt, it.ct});
          if (side == 0 && at_ == tile_at.end()) throw ConpError(CONP_ERR_STATE, "sk_gemm schedule: a band starts in a row tile without work");
          if (at_ == tile_at.end() || (side == 1 && !any)) continue;
          of_tile[at_->second].push_back({(int)sgm, side});
        }
      }
      int slot = 0;
      max_nsplit = 0;
      for (size_t i = 0; i < tiles_h.size(); ++i) {
        tiles_h[i].item0 = slot; tiles_h[i].nsplit = (int)of_tile[i].size();
        for (const auto &e : of_tile[i]) (e.second ? seg_sgb : seg_sga)[e.first] = slot++;
        max_nsplit = std::max(max_nsplit, tiles_h[i].nsplit);
      }
      n_slots = slot;
    }
    d_items.upload(items_h, stream);
    d_seg_ptr.upload(seg_ptr_h, stream);
    d_seg_idx.upload(seg_idx_h, stream);
    // the kernel's own work list: one fixed-size row per workgroup (SkWItem)
    witems_maxseg = 1;
    for (int w = 0; w < nwg; ++w) witems_maxseg = std::max(witems_maxseg, (int)per_wg[w].size());
    std::vector<int> slab_slot_of(items_h.size(), -1);
    n_slab_slots = 0; max_seg_chunks = 0;
    for (size_t sgm = 0; sgm < items_h.size(); ++sgm) {          // items are band-major, chunk-ordered inside a band
      max_seg_chunks = std::max(max_seg_chunks, items_h[sgm].c1 - items_h[sgm].c0);
      if (items_h[sgm].g0 == 0 && items_h[sgm].ct == 0) slab_slot_of[sgm] = n_slab_slots++;
    }
    std::vector<SkWItem> wl((size_t)nwg * witems_maxseg, SkWItem{0, 4, 0, 0, 0, 0, 0, -1, 0, -1, 0ull});
    for (int w = 0; w < nwg; ++w)
      for (size_t k = 0; k < per_wg[w].size(); ++k) {
        const int sg = per_wg[w][k];
        const SkItem &it = items_h[sg];
        // (SkFuse: the segments of band 0 of column tile 0 tile the atoms once: they carry the slab term's sum of q z, in chunk order)
        wl[(size_t)w * witems_maxseg + k] = SkWItem{it.g0, it.rf, it.ct, it.c0, it.c1, sg, seg_sga[sg], seg_sgb[sg], k == 0 ? (int)per_wg[w].size() : 0,
                                                    slab_slot_of[sg], it.nbf};
      }
    d_witems.upload(wl, stream);
    d_tiles.upload(tiles_h, stream);
    // projecting mode: a segment leaves ONE band-local piece (slot = its index; a band's segments are contiguous).  hc_sum_kernel
    // adds a band's pieces and maps its rows to the plan's; when every band IS a row tile of the plan (rf = 4, aligned: the decks,
    // CONP_SK_BANDS4) the dot kernel may add a row tile's few pieces itself, from the per-row-tile lists below
    bands_aligned = true;
    for (const Band &bd : bands) if (bd.rf != 4 || (bd.g0 & 3)) bands_aligned = false;
    {
      // per row fragment of the plan: the pieces that hold it, column tile after column tile, segment after segment (the order
      // the sums are formed in: fixed by the schedule, not by which workgroup finishes first)
      n_frags = 4 * plan.n_row_tiles;
      std::vector<std::vector<int2>> of_frag(n_frags);
      for (size_t sgm = 0; sgm < items_h.size(); ++sgm) {
        const SkItem &it = items_h[sgm];
        for (int f = 0; f < it.rf; ++f)
          if (it.g0 + f < n_frags && ((it.nbf >> (8 * f)) & 255u) != 0)
            of_frag[it.g0 + f].push_back(make_int2((int)(sgm * (size_t)sk_hc_stride()) + 16 * f, it.rf));
      }
      std::vector<int> fptr(n_frags + 1, 0);
      std::vector<int2> fent;
      for (int g = 0; g < n_frags; ++g) { fent.insert(fent.end(), of_frag[g].begin(), of_frag[g].end()); fptr[g + 1] = (int)fent.size(); }
      if (fent.empty()) fent.push_back(make_int2(0, 4));
      if (items_h.size() * (size_t)sk_hc_stride() > 0x7fffffffull) throw ConpError(CONP_ERR_STATE, "sk_gemm schedule: piece offsets exceed 31 bits");
      d_frag_ptr.upload(fptr, stream); d_frag_ents.upload(fent, stream);
    }
    std::vector<int> hptr(own_rt_h.size() + 1, 0), hidx;
    for (size_t k = 0; k < own_rt_h.size(); ++k) {
      if (bands_aligned)
        for (size_t sgm = 0; sgm < items_h.size(); ++sgm) if ((items_h[sgm].g0 >> 2) == own_rt_h[k]) hidx.push_back((int)sgm);
      hptr[k + 1] = (int)hidx.size();
    }
    hslots = bands_aligned ? (int)hidx.size() : (int)items_h.size();
    if (hidx.empty()) hidx.push_back(0);
    d_hslot_ptr.upload(hptr, stream); d_hslot_idx.upload(hidx, stream);
  }
  int n_slab_slots = 0, max_seg_chunks = 0;      // SkFuse: segments that carry the slab sum; the longest segment (chunks)
  bool bands_aligned = true;       // every band of sk_gemm's schedule is a row tile of the plan
  int n_frags = 0;
  DevBuf<int> d_frag_ptr;
  DevBuf<int2> d_frag_ents;

  void gather_xele(const conp_atoms *at) {
    const int ne = idx.elenum_all;
    xele_h.assign((size_t)ne * 3, 0.0);
    std::vector<double> mine((size_t)std::max(idx.elenum, 1) * 3);
    for (int i = 0; i < idx.elenum; ++i) {
      const int iloc = idx.tag2local[idx.ele2tag[i]];
      for (int c = 0; c < 3; ++c) mine[3 * (size_t)i + c] = at->x[3 * (size_t)iloc + c];
    }
    ele_comm(mine.data(), 3, xele_h.data());           // km_ewald.cpp:510-531 gathers the electrode tables the same way
  }

  // km_ewald.cpp:134-145 a_read: electrode phase tables (electrodes are immobile: filled once, appendix D).
  // sincos_a_ele / sincos_a_comm_eleall (km_ewald.cpp:426-531) on the device (conp_tables.hip): the host computes the 3 Ne seed pairs
  // (cos, sin)(unitk_c x_ic) with libm (conp_host.cpp electrode_seeds: the reference's own calls, every table entry keeps its bits) and the
  // kernels run the recurrences and the (kx, +-ky) products straight into Xe / Ye / Tz / Rp.
  bool tables_current = false;   // false after km_conp_setup re-planned: tables, z classes and the projection block are rebuilt
  DevBuf<double> d_seeds;
  DevBuf<int> d_zrep;
  void km_a_read(const conp_atoms *at) {
    const int ne = idx.elenum_all;
    gather_xele(at);
    // (function scope: the uploads are asynchronous, the vectors must live until the sync() at the end)
    std::vector<double> seeds, z(ne_pad, 0.0);
    std::vector<int> zclass(ne_pad, 0), rep;
    electrode_seeds(kt, ne, xele_h.data(), seeds);        // km_ewald.cpp:440-442, compiled like the reference (conp_host.cpp)
    for (int i = 0; i < ne; ++i) z[i] = xele_h[3 * (size_t)i + 2];
    d_seeds.upload(seeds, stream); d_ele_z.upload(z, stream);
    // padding rows / atoms stay zero (row kxmax + 1 of Xe: padding planar vectors point there)
    d_Xe.reserve((size_t)(plan.kxmax + 2) * ne_pad); d_Ye.reserve((size_t)(plan.kymax + 1) * ne_pad);
    d_Rp.reserve((size_t)plan.R_pad * ne_pad); d_Tz.reserve((size_t)plan.C_pad * ne_pad);
    HIP_TRY(hipMemsetAsync(d_Xe.p, 0, (size_t)(plan.kxmax + 2) * ne_pad * sizeof(double2), stream));
    HIP_TRY(hipMemsetAsync(d_Ye.p, 0, (size_t)(plan.kymax + 1) * ne_pad * sizeof(double2), stream));
    HIP_TRY(hipMemsetAsync(d_Rp.p, 0, (size_t)plan.R_pad * ne_pad * sizeof(double), stream));
    HIP_TRY(hipMemsetAsync(d_Tz.p, 0, (size_t)plan.C_pad * ne_pad * sizeof(double), stream));
    launch_ele_tables(stream, dplan, plan.kzt, ne, ne_pad, d_seeds.p, d_Xe.p, d_Ye.p, d_Tz.p, d_Rp.p);
    csk_h.clear(); snk_h.clear();                    // conp_fix_get_ele_trig reads the device tables back on request
    if (args.pppm) {   // aaa_map_rho (pppm_conp.cpp:318-344): stencil weights and lower-left mesh index of every electrode atom
      std::vector<int> eg((size_t)ne * 3);
      std::vector<double> ew((size_t)ne * 24, 0.0);
      for (int i = 0; i < ne; ++i)
        for (int c = 0; c < 3; ++c) {
          const double xlo = xele_h[3 * (size_t)i + c] - pppm.boxlo[c];
          const int n = static_cast<int>(xlo * pppm.delinv[c] + pppm.shift) - PppmPlan::OFFSET;
          eg[3 * (size_t)i + c] = n;
          pppm.rho1d(n + pppm.shiftone - xlo * pppm.delinv[c], &ew[(size_t)i * 24 + 8 * c]);
        }
      d_pp_egrid.upload(eg, stream); d_pp_ew.upload(ew, stream);
    }
    // z classes: atoms with bitwise equal z share their Tz column
    {
      std::map<double, int> cls;
      for (int i = 0; i < ne; ++i) {
        auto it = cls.find(z[i]);
        if (it == cls.end()) { it = cls.emplace(z[i], (int)cls.size()).first; rep.push_back(i); }
        zclass[i] = it->second;
      }
      const bool off = diag_switch("CONP_NO_ZCLASS") != nullptr;
      nzc = (!off && cls.size() <= 64 && 4 * cls.size() <= (size_t)ne) ? (int)cls.size() : 0;   // worthwhile only if it compresses
      // b_zc_dot keeps Hc for 32 rows of every row tile and every class in LDS
      if ((size_t)plan.n_row_tiles * 32 * (size_t)nzc * sizeof(double) > 96 * 1024) nzc = 0;
      if (nzc > 0) {
        d_zclass.upload(zclass, stream); d_zrep.upload(rep, stream);
        d_Tzc.reserve((size_t)plan.C_pad * 64); d_TzcT.reserve((size_t)std::max(nzc, 1) * plan.C_pad);
        d_Tzc.zero(stream); d_TzcT.zero(stream);
        // Tzc[t][c] = Tz[t][representative of class c]; class-major copy for sk_gemm's projecting epilogue
        launch_ele_zclass(stream, plan.C_pad, ne_pad, nzc, d_zrep.p, d_Tz.p, d_Tzc.p, d_TzcT.p);
        skproj_h = SkProj{d_wfull.p, d_TzcT.p, nzc, plan.C_pad};      // (a member: the asynchronous upload reads it after this returns)
        d_skproj.upload(&skproj_h, 1, stream);
        d_Hc.reserve((size_t)8 * plan.R_pad * 64); d_Hc.zero(stream);     // 8 slots: the reduction's column slices (b_hc: 4, the rest stay 0)
      }
      // the stream-K schedule was cut before the electrodes' geometry was known: planar electrodes take more, smaller shares
      if (sk_projects() && !tiles_h.empty() && nl_pad > 0) { build_items(); reserve_partials(); }
    }
    sync();
    tables_current = true;
    zn_tables_current = false;      // the z-window table P depends on the weights and the z-class phases
  }

  // Library-owned page-locked staging for the host-buffer hooks: the charges and scalars come back in ONE place with one
  // stream synchronisation (a hipMemcpyAsync into pageable memory blocks per call), and small x / q uploads go through it so
  // that they are real asynchronous DMA transfers.
  double *h_pin = nullptr;
  size_t h_pin_n = 0;
  double *pinned(size_t n) {
    if (n > h_pin_n) {
      if (h_pin) { sync(); (void)hipHostFree(h_pin); h_pin = nullptr; h_pin_n = 0; }
      const size_t want = n + n / 4 + 64;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_pin), want * sizeof(double), hipHostMallocDefault));
      h_pin_n = want;
    }
    return h_pin;
  }
  // conp_env.ghost_images: which owned atom and which periodic image every ghost is (verified here, at a re-neighbour)
  bool ghost_mode = false;
  DevBuf<int> d_ghost_owner, d_ghost_img;
  void map_ghosts(const conp_atoms *at) {
    ghost_mode = false;
    if (!env.ghost_images || at->nghost <= 0) return;
    const int nl_ = at->nlocal, ng = at->nghost;
    const double prd[3] = {env.xprd, env.yprd, env.zprd};
    std::vector<int> owner(ng), img((size_t)ng * 3);
    for (int g = 0; g < ng; ++g) {
      const int i = nl_ + g, t = at->tag[i];
      const int o = (t >= 0 && t < (int)idx.tag2local.size()) ? idx.tag2local[t] : -1;
      if (o < 0 || at->q[i] != at->q[o]) return;                       // not an image of an owned atom: full uploads
      for (int c = 0; c < 3; ++c) {
        const double xo = at->x[3 * (size_t)o + c], xg = at->x[3 * (size_t)i + c];
        const int n = (int)std::lround((xg - xo) / prd[c]);
        const double shift = n * prd[c];
        if (xo + shift != xg) return;
        img[3 * (size_t)g + c] = n;
      }
      owner[g] = o;
    }
    d_ghost_owner.upload(owner, stream); d_ghost_img.upload(img, stream);
    ghost_mode = true;
  }

  static constexpr size_t STAGE_MAX = 1u << 16;      // doubles: above 512 KB the pageable copy pipelines better than memcpy + DMA
  void upload_xq(const conp_atoms *at) {
    const int nup = ghost_mode ? at->nlocal : nall;    // ghost_images: owned atoms only, ghosts rebuilt on the device
    upload_xq_n(at, nup);
    if (ghost_mode)
      launch_ghost_fill(stream, at->nlocal, at->nghost, d_ghost_owner.p, d_ghost_img.p, env.xprd, env.yprd, env.zprd, d_x.p, d_q.p);
  }
  // conp_fix_pin_host_arrays: the host's own x / q arrays, page-locked in place (asynchronous DMA straight out of them)
  const double *pin_x = nullptr, *pin_q = nullptr;
  int pin_n = 0;
  void unpin_host() {
    if (pin_x || pin_q) sync();                       // no transfer out of them may be in flight
    if (pin_x) (void)hipHostUnregister(const_cast<double *>(pin_x));
    if (pin_q) (void)hipHostUnregister(const_cast<double *>(pin_q));
    pin_x = pin_q = nullptr; pin_n = 0;
  }
  void pin_host(const double *x, const double *q, int n) {
    if (x == pin_x && q == pin_q && n <= pin_n) return;
    unpin_host();
    if (!x || !q || n <= 0) return;
    if (hipHostRegister(const_cast<double *>(x), (size_t)n * 3 * sizeof(double), hipHostRegisterDefault) != hipSuccess) {
      (void)hipGetLastError();
      throw ConpError(CONP_ERR_NO_DEVICE, "conp_fix_pin_host_arrays: the runtime refused to page-lock x (the staged copy stays in use)");
    }
    if (hipHostRegister(const_cast<double *>(q), (size_t)n * sizeof(double), hipHostRegisterDefault) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipHostUnregister(const_cast<double *>(x));
      throw ConpError(CONP_ERR_NO_DEVICE, "conp_fix_pin_host_arrays: the runtime refused to page-lock q (the staged copy stays in use)");
    }
    pin_x = x; pin_q = q; pin_n = n;
  }
  void upload_xq_n(const conp_atoms *at, int nup) {
    const size_t nx = (size_t)nup * 3, nq = (size_t)nup;
    bool pinned_ok = at->x == pin_x && at->q == pin_q && nup <= pin_n;
    if (pinned_ok) {
      // pointer equality alone does not prove the registration is alive: a host may have freed the arrays and been handed the same
      // addresses again (create_atoms / read_dump between runs).  The runtime knows (a sub-microsecond table look-up).
      hipPointerAttribute_t pa;
      if (hipPointerGetAttributes(&pa, at->x) != hipSuccess || pa.type != hipMemoryTypeHost) { (void)hipGetLastError(); pinned_ok = false; }
      else if (hipPointerGetAttributes(&pa, at->q) != hipSuccess || pa.type != hipMemoryTypeHost) { (void)hipGetLastError(); pinned_ok = false; }
      if (!pinned_ok) { pin_x = pin_q = nullptr; pin_n = 0; }      // (nothing to unregister: the registration went with the memory)
    }
    if (pinned_ok) {
      // page-locked by the host: two asynchronous DMA transfers, nothing staged, nothing blocks
      HIP_TRY(hipMemcpyAsync(d_x.p, at->x, nx * sizeof(double), hipMemcpyHostToDevice, stream));
      HIP_TRY(hipMemcpyAsync(d_q.p, at->q, nq * sizeof(double), hipMemcpyHostToDevice, stream));
    } else if (nx + nq <= STAGE_MAX) {
      // results live in the first ne_pad + 8 doubles (update_charge); uploads behind them
      double *st = pinned((size_t)ne_pad + 8 + nx + nq) + ne_pad + 8;
      sync();                                          // the previous update's transfers out of the staging area are done
      std::memcpy(st, at->x, nx * sizeof(double));
      std::memcpy(st + nx, at->q, nq * sizeof(double));
      HIP_TRY(hipMemcpyAsync(d_x.p, st, nx * sizeof(double), hipMemcpyHostToDevice, stream));
      HIP_TRY(hipMemcpyAsync(d_q.p, st + nx, nq * sizeof(double), hipMemcpyHostToDevice, stream));
    } else {
      HIP_TRY(hipMemcpyAsync(d_x.p, at->x, nx * sizeof(double), hipMemcpyHostToDevice, stream));
      HIP_TRY(hipMemcpyAsync(d_q.p, at->q, nq * sizeof(double), hipMemcpyHostToDevice, stream));
    }
  }

  // km_ewald.cpp:147-151 + :584-666 : k-space part of A into d_A (strict lower triangle + diagonal + slab on j <= i)
  // several ranks with a way to sum matrices (RCCL or the host callbacks): the tiles are dealt to the ranks
  // (an RCCL communicator of ONE rank takes the same path: that is how the call sites are tested on a one-GPU box)
  bool setup_sharded() const { return nccl != nullptr || (env.nranks > 1 && rc.active()); }
  // sum of the ranks' zero-initialised partial matrices (fix_conp.cpp:816-822 gathers rows instead; every element here is
  // written by one rank's k-space tile and at most one rank's real-space row, so the sum does not depend on the order)
  void allreduce_matrix(double *A, size_t n) {
    if (!setup_sharded()) return;
    if (nccl) {
      g_rccl.ok(g_rccl.AllReduce(A, A, n, ncclDouble, ncclSum, nccl, stream), "all-reduce(A)");
      return;
    }
    std::vector<double> h(n);
    HIP_TRY(hipMemcpyAsync(h.data(), A, n * sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    rc.sum(h.data(), (int64_t)n);
    HIP_TRY(hipMemcpyAsync(A, h.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
    sync();
  }
  void km_a_cal_device() {
    const int ne = idx.elenum_all;
    const int trank = setup_sharded() ? env.rank : 0, tranks = setup_sharded() ? env.nranks : 1;
    // planar electrodes (z classes, as in the projection's fast path): contraction over the planar rows only -- ~100x fewer flops.
    // CONP_A_GENERAL: comparison switch, always the (planar, kz) contraction.
    if (nzc > 0 && nzc <= 8 && !path_on(CONP_PATH_A_GENERAL)) {
      d_A.reserve((size_t)ne * ne);
      d_A.zero(stream);
      d_Wz.reserve((size_t)plan.R_pad * nzc * nzc);
      prof.begin("a_kspace", stream);
      launch_a_kspace_zclass(stream, dplan, ne, ne_pad, nzc, d_Rp.p, d_Tzc.p, d_zclass.p, d_Wz.p, d_A.p, trank, tranks);
      prof.end(stream);
      return;
    }
    // kz chunks (16-kz blocks) dealt to nsplit groups of about equal work: heaviest first to the lightest group; a chunk costs
    // as many row tiles as reach it
    const int nchunk = plan.n_col_tiles * KPlan::CT_BLK;
    std::vector<int> group(nchunk, -1);
    int nsplit = 1;
    {
      std::vector<std::pair<int, int>> cost;
      for (int c = 0; c < nchunk; ++c) {
        int n = 0;
        for (int rt = 0; rt < plan.n_row_tiles; ++rt) n += plan.nba(rt, c / KPlan::CT_BLK) > c % KPlan::CT_BLK;
        if (n) cost.push_back({-n, c});
      }
      nsplit = a_kspace_nsplit(ne_pad, num_cus, (int)cost.size(), tranks);
      std::stable_sort(cost.begin(), cost.end());
      std::vector<int> load(nsplit, 0);
      for (auto &e : cost) {
        int g = 0;
        for (int k = 1; k < nsplit; ++k) if (load[k] < load[g]) g = k;
        group[e.second] = g; load[g] += -e.first;
      }
    }
    d_a_chunk_group.upload(group, stream);
    d_A.reserve((size_t)nsplit * ne * ne);
    d_A.zero(stream);
    prof.begin("a_kspace", stream);
    launch_a_kspace(stream, dplan, ne, ne_pad, d_Rp.p, d_Tz.p, d_A.p, nsplit, d_a_chunk_group.p, trank, tranks);
    prof.end(stream);
  }

  // fix_conp.cpp:777-861 a_cal
  void a_cal(const conp_atoms *at) {
    if (!have_alist) throw ConpError(CONP_ERR_STATE, "a_cal: no electrode neighbor list (init_list not called)");
    const int ne = idx.elenum_all;
    logf("A matrix calculating ...\n");                                   // :787
    const auto t_a0 = std::chrono::steady_clock::now();
    static const bool tt = getenv("CONP_TIME_REN") != nullptr;
    auto tm = std::chrono::steady_clock::now();
    auto mk = [&](const char *what) {
      if (!tt) return;
      sync();
      const auto t = std::chrono::steady_clock::now();
      std::fprintf(stderr, "    a_cal %-20s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tm).count());
      tm = t;
    };
    ++s_generation;
    km_a_read(at);
    mk("electrode tables");
    km_a_cal_device();
    mk("k-space SYRK");
    const double MY_PIS = 1.77245385090551602729;
    const double diag_k = kt.ug_tot - (2.0 / MY_PIS) * kt.g_ewald;      // km_ewald.cpp:631-634
    const double diag_self = (std::sqrt(2.0) / MY_PIS) * args.eta;      // fix_conp.cpp:796-801
    const double pref = 12.56637061435917295384 / kt.volume;            // MY_4PI/volume km_ewald.cpp:648
    const double *diag_atom = nullptr;
    if (ehgo_active) {            // :803-810: u0 of the atom's type on the diagonal
      std::vector<double> da(ne_pad, 0.0), mine(std::max(idx.elenum, 1));
      for (int i = 0; i < idx.elenum; ++i) mine[i] = u0_i_h[at->type[idx.tag2local[idx.ele2tag[i]]]];
      ele_comm(mine.data(), 1, da.data());
      d_diag_atom.upload(da, stream);
      diag_atom = d_diag_atom.p;
    }
    const bool sharded = setup_sharded();
    if (!sharded || env.rank == 0)         // diagonal and slab term once (they overwrite / add on the whole lower triangle)
      launch_a_diag_slab(stream, ne, diag_k, diag_self, diag_atom, kt.slabflag == 1, pref, d_ele_z.p, d_A.p);
    build_a_rows(alist, at->nlocal, at->tag, at->echeck, idx, env.newton_pair != 0, arows);
    d_a_rowptr.upload(arows.row_ptr, stream); d_a_ele.upload(arows.ele_atom, stream);
    d_a_oth.upload(arows.oth_atom, stream); d_a_col.upload(arows.col, stream);
    upload_xq(at);
    prof.begin("a_real", stream);
    // real-space rows: a sub-domain's list only holds the rows of its own electrode atoms (fix_conp.cpp:1242-1276); with
    // replicated atoms the rows are split by the row range
    const bool by_range = sharded && !decomposed;
    launch_a_real(stream, ne, by_range ? row0 : 0, by_range ? row1 : ne, d_a_rowptr.p, d_a_ele.p, d_a_oth.p, d_a_col.p, d_x.p,
                  d_type.p, real_params(), d_A.p);
    prof.end(stream);
    allreduce_matrix(d_A.p, (size_t)ne * ne);
    launch_a_symmetrise(stream, ne, d_A.p);
    HIP_TRY(hipGetLastError());
    sync();
    mk("real space + symmetrise");
    runstage = 1;
    if (args.matout) write_matrix_file("amatrix", 0);          // fix_conp.cpp:833-849
    logf("A matrix calculation time  = %g\n",                               // :857
         std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a0).count());
  }

  // fix_conp.cpp:721-773 a_read: `org F` / `inv F`
  void a_read_file(const conp_atoms *at, const char *path) {
    const int ne = idx.elenum_all;
    FILE *fp = std::fopen(path, "r");
    if (!fp) throw ConpError(CONP_ERR_IO, "Invalid fix conp command (Cannot open A matrix file)");   // :143
    std::vector<int> tags;
    std::vector<double> a((size_t)ne * ne);
    size_t count = 0;
    const size_t need = (size_t)ne + (size_t)ne * ne;
    char tok[64];
    bool too_many = false;
    while (std::fscanf(fp, "%63s", tok) == 1) {
      if (count < (size_t)ne) tags.push_back(std::atoi(tok));
      else if (count < need) a[count - ne] = std::atof(tok);
      else { too_many = true; break; }
      ++count;
    }
    std::fclose(fp);
    if (too_many) throw ConpError(CONP_ERR_IO, "Too many entries in A matrix file");        // :737
    if (count != need) throw ConpError(CONP_ERR_IO, "Too few entries in A matrix file");     // :746
    try { idx.renumber_from_tags(tags, at->nlocal, at->tag, at->echeck, &rc); }
    catch (const std::exception &e) { throw ConpError(CONP_ERR_IO, e.what()); }
    upload_atoms_static(at);                                   // atom2eleall follows the new numbering
    build_b_rows_device(at);                                   // rows follow the new numbering (the list is on the device)
    ++s_generation;
    HIP_TRY(hipMemcpyAsync(d_A.p, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice, stream));
    km_a_read(at);                                             // kspmod->a_read(): electrode phase tables (:772)
    sync();
    matrix_loaded = true;
    runstage = 1;
  }

  // fix_conp.cpp:833-849 (amatrix) and :960-977 (inv_a_matrix)
  void write_matrix_file(const char *path, int which) {
    const int ne = idx.elenum_all;
    unshard_rows();
    std::vector<double> a((size_t)ne * ne);
    HIP_TRY(hipMemcpyAsync(a.data(), d_A.p, a.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    FILE *fp = std::fopen(path, "w");
    if (!fp) throw ConpError(CONP_ERR_IO, std::string("cannot open ") + path + " for writing");
    if (which == 0) {
      std::fprintf(fp, " ");
      for (int i = 0; i < ne; ++i) std::fprintf(fp, "%20d", idx.eleall2tag[i]);
      std::fprintf(fp, "\n");
      for (int i = 0; i < ne; ++i) {
        std::fprintf(fp, " ");
        for (int j = 0; j < ne; ++j) std::fprintf(fp, "%20.12f", a[(size_t)i * ne + j]);
        std::fprintf(fp, "\n");
      }
    } else {
      for (int i = 0; i < ne; ++i) { if (i == 0) std::fprintf(fp, " "); std::fprintf(fp, "%20d", idx.eleall2tag[i]); }
      std::fprintf(fp, "\n");
      for (size_t k = 0; k < a.size(); ++k) {
        if (k % ne != 0) std::fprintf(fp, " ");
        std::fprintf(fp, "%20.10f", a[k]);
        if ((k + 1) % ne == 0) std::fprintf(fp, "\n");
      }
    }
    std::fclose(fp);
  }

  // fix_conp.cpp:609-637 b_setq_cal (host: Ne scalars)
  void b_setq_cal(const conp_atoms *at) {
    const int ne = idx.elenum_all;
    const double zprd = env.zprd, zhalf = 0.5 * env.zprd + env.boxlo_z;
    d_vec_h.assign(ne_pad, 0.0);
    std::fill(idx.elecheck_eleall.begin(), idx.elecheck_eleall.end(), 0);
    std::vector<double> mine((size_t)std::max(idx.elenum, 1) * 2), all((size_t)std::max(ne, 1) * 2);
    for (int iloc = 0; iloc < idx.elenum; ++iloc) {
      const int i = idx.tag2local[idx.ele2tag[iloc]];
      const int eci = at->echeck[i];
      const double z = at->x[3 * (size_t)i + 2];
      double v;
      if (args.ff_flag == CONP_FF_FFIELD) {
        if (eci == 1 && z < zhalf) v = -evscale * (z / zprd + 1);
        else v = -evscale * z / zprd;
      } else v = -0.5 * evscale * eci;
      mine[2 * (size_t)iloc] = v; mine[2 * (size_t)iloc + 1] = (double)eci;
    }
    ele_comm(mine.data(), 2, all.data());            // b_comm(bbb, bbb_all) :634 and the Allreduce of elecheck_eleall :633
    for (int i = 0; i < ne; ++i) { d_vec_h[i] = all[2 * (size_t)i]; idx.elecheck_eleall[i] = (int)all[2 * (size_t)i + 1]; }
    std::vector<int> ec(ne_pad, 0);
    for (int i = 0; i < ne; ++i) ec[i] = idx.elecheck_eleall[i];
    d_elecheck.upload(ec, stream);
    HIP_TRY(hipMemcpyAsync(d_b, d_vec_h.data(), ne * sizeof(double), hipMemcpyHostToDevice, stream));
    if (args.cond) {               // FixCond::cond_setup (fix_cond.cpp:46-56): z-hat vector = preset vector / evscale
      std::vector<double> sz(ne_pad, 0.0);
      for (int i = 0; i < ne; ++i) sz[i] = d_vec_h[i] / evscale;
      d_setzvec.upload(sz, stream);
      cond_ready = false;
    }
    sync();
    if (runstage == 1) runstage = 2;
  }

  // fix_conp.cpp:982-1067 inv_project on device, bit-exact operation order
  void inv_project_device(int n, double *A, const std::vector<double> *eleallz, double zhalf) {
    ++s_generation;                          // (every writer of the device matrix says so: the packed copy follows)
    d_ainve.reserve(n);
    launch_inv_project(stream, n, A, 0, nullptr, d_ainve.p, d_scalars.p + 4, 0);
    HIP_TRY(hipMemcpyAsync(&totinve, d_scalars.p + 4, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    mesgf("conp output: <e,e> = %.8g\n", totinve * evscale);             // :1006-1009
    if (args.nullneutral) {
      launch_inv_project_apply(stream, n, A, d_ainve.p, d_scalars.p + 4);
      if (args.zneutr) {
        std::vector<unsigned char> mask(n, 0);
        for (int i = 0; i < n; ++i) mask[i] = (*eleallz)[i] > zhalf;
        d_mask.upload(mask, stream);
        launch_inv_project(stream, n, A, 1, d_mask.p, d_ainve.p, d_scalars.p + 5, 1);
      }
    }
    sync();
  }

  void inv_project() {
    const int ne = idx.elenum_all;
    std::vector<double> z(ne);
    for (int i = 0; i < ne; ++i) z[i] = xele_h[3 * (size_t)i + 2];
    inv_project_device(ne, d_A.p, &z, 0.5 * env.zprd + env.boxlo_z);
  }

  // LU-quality inverse in place of dgetrf_/dgetri_ (fix_conp.cpp:947-949): blocked Gauss-Jordan, partial pivoting
  void invert_device(int n, double *A) {
    ++s_generation;
    d_inv_work.reserve(inverse_workspace_doubles(n));
    d_ipiv.reserve(n + 1); d_info.reserve(2);
    // the in-place elimination destroys A: keep a copy while the multi-workgroup panel (grid barriers) is in use, so that a
    // barrier time-out (info = -7: some workgroup was not resident) can be answered by the one-workgroup panel
    // comparison / test switches, read per call and only here: CONP_PANEL_SINGLE = the one-workgroup panel, CONP_PANEL_MAXG = cap
    // on the panel's workgroups, CONP_PANEL_SPIN = polls of its grid barrier before it gives up (0 forces the time-out: the
    // restore-and-retry below runs, tests/test_gpu_parity.py)
    const bool single = env_knob("CONP_PANEL_SINGLE") != nullptr;
    const int max_wg = env_knob("CONP_PANEL_MAXG") ? atoi(env_knob("CONP_PANEL_MAXG")) : 0;
    const unsigned spin_limit = env_knob("CONP_PANEL_SPIN") ? (unsigned)strtoul(env_knob("CONP_PANEL_SPIN"), nullptr, 10) : (1u << 22);
    inverse_retries = 0;
    inverse_path = 2;
    // CONP_INV_GENERAL: comparison switch, always the pivoted elimination
    const bool try_spd = !path_on(CONP_PATH_INV_PIVOTED);
    d_inv_backup.reserve((size_t)n * n);
    HIP_TRY(hipMemcpyAsync(d_inv_backup.p, A, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    int info = 0;
    if (try_spd) {
      // an exactly symmetric matrix (the fix's own A always is, fix_conp.cpp:826-831) is first taken for positive definite: block
      // Gauss-Jordan on its own diagonal, no pivot search.  A non-positive pivot says it was not (info = -8): restore, pivot.
      int sym = 0;
      launch_symmetry_check(stream, n, A, d_info.p);
      HIP_TRY(hipMemcpyAsync(&sym, d_info.p, sizeof(int), hipMemcpyDeviceToHost, stream));
      sync();
      if (sym) {
        prof.begin("inverse", stream);
        launch_inverse_spd(stream, n, A, d_inv_work.p, d_info.p);
        prof.end(stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(&info, d_info.p, sizeof(int), hipMemcpyDeviceToHost, stream));
        sync();
        if (info == 0) { inverse_path = 1; d_inv_backup.release(); return; }
        HIP_TRY(hipMemcpyAsync(A, d_inv_backup.p, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, stream));
        info = 0;
      }
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
      prof.begin("inverse", stream);
      const bool multi = launch_inverse(stream, n, A, d_inv_work.p, d_ipiv.p, d_info.p, num_cus, attempt == 0 && !single, max_wg, spin_limit);
      prof.end(stream);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(&info, d_info.p, sizeof(int), hipMemcpyDeviceToHost, stream));
      sync();
      if (!(multi && info == -7)) break;
      ++inverse_retries;
      mesgf("conp/hip: the inverse's multi-workgroup panel timed out at its grid barrier; repeating with the one-workgroup panel\n");
      HIP_TRY(hipMemcpyAsync(A, d_inv_backup.p, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, stream));
    }
    d_inv_backup.release();
    if (info != 0) throw ConpError(CONP_ERR_NUMERIC, "Inversion failed!");   // fix_conp.cpp:956
  }

  // fix_conp.cpp:932-980 inv
  void inv() {
    if (runstage == 2 && args.a_matrix_f < 2) {
      const int ne = idx.elenum_all;
      invert_device(ne, d_A.p);
      if (!env.one_electrode) inv_project();
      if (args.matout) write_matrix_file("inv_a_matrix", 1);   // :960-977
    }
    if (runstage == 2) runstage = 3;
  }

  // fix_conp.cpp:864-930 cg
  // The iterations are enqueued in batches and the convergence flag is read back once per batch (every kernel of an iteration
  // after convergence returns at once).  The first batch is as long as the last solve needed, so a typical update costs one
  // read-back: scalars, flag, net charge and the residual history come over in ONE copy into the page-locked staging buffer.
  int cg_batch = 8;
  const bool cg_unfused = path_on(CONP_PATH_CG_TWO_LAUNCH);      // comparison switch: two launches per iteration (round 1)
  bool cg_persist_off = false;
  long cg_solves = 0;
  DevBuf<unsigned> d_cg_tk;            // two grid-barrier words of the one-launch solve, alternating between solves
  void cg() {
    const int ne = idx.elenum_all;
    const int nctl = 16 + args.maxiter + 1;                  // scal[0..12], pad, hist[0..maxiter] at offset 16
    d_cg_res.reserve((size_t)2 * ne); d_cg_p.reserve((size_t)2 * ne); d_cg_ap.reserve((size_t)2 * ne);
    d_cg_scal.reserve(nctl); d_cg_done.reserve(1);
    const bool two_launch = cg_unfused || !cg_step_fits(ne) || args.maxiter <= 1;
    if (two_launch) d_cg_done.zero(stream);                  // (the one-launch form's start kernel clears the flag itself)
    double *hist_dev = d_cg_scal.p + 16;
    double *ctl = pinned((size_t)ne_pad + 8 + nctl) + ne_pad + 8;
    int done = 0, iter = 1, batch = std::max(2, std::min(16, cg_batch));
    prof.begin("cg", stream);
    // round 5, measured and NOT the default (CONP_PATH_CG_PERSIST selects it; profiles/r05_cg_persist_ab.txt): the whole solve as ONE
    // persistent launch -- the matrix stays in the XCDs' L2s between iterations, which a kernel boundary invalidates (cg_step_kernel
    // fetches 8 n^2 bytes from the memory side at every launch) -- with one fence-free grid barrier per iteration.  il_twolayer: 92-94 us
    // per solve against 76-78 with a launch per iteration: the barrier (write-through of the products, ticket, poll, sc1 read-back of the
    // product vector by every workgroup) costs ~2.5 us more than a kernel boundary plus the re-read of 22 MB from the Infinity Cache.
    bool persisted = false;
    if (!two_launch && !cg_persist_off && cg_persist_fits(ne) && !results_by_copy && path_on(CONP_PATH_CG_PERSIST)) {
      if (d_cg_tk.n == 0) { d_cg_tk.reserve(2); d_cg_tk.zero(stream); }
      unsigned *tk = d_cg_tk.p + (cg_solves & 1), *tkn = d_cg_tk.p + ((cg_solves + 1) & 1);
      if (launch_cg_persist(stream, num_cus, ne, d_A.p, d_b, d_eleallq, d_cg_ap.p, d_cg_scal.p, args.tolerance, args.maxiter, hist_dev, ctl,
                            tk, tkn, 1u << 20)) {
        ++cg_solves;
        if (spec_dq) scatter_device(spec_dq, spec_pot, false);      // update_direct: the charge write rides behind the solve
        sync();
        if (ctl[8] < 0.0) {
          // a workgroup's wait at the grid barrier ran out (the workgroups were not all resident: another kernel on the device?):
          // this handle takes a launch per iteration from now on; the solve is repeated (b is untouched, q is reset by the start)
          cg_persist_off = true;
          mesgf("conp/hip: the one-launch CG solve timed out at its grid barrier; falling back to one launch per iteration\n");
        } else {
          persisted = true;
          done = ctl[8] != 0.0;
          if (spec_dq && done) spec_done = true;
        }
      }
    }
    if (persisted) {
    } else if (two_launch) {
      launch_cg_init(stream, ne, d_A.p, d_b, d_eleallq, d_cg_res.p, d_cg_p.p, d_cg_scal.p);
      while (iter < args.maxiter && !done) {
        const int batch_end = std::min(args.maxiter, iter + batch);
        for (; iter < batch_end; ++iter)
          launch_cg_iter(stream, ne, d_A.p, d_eleallq, d_cg_res.p, d_cg_p.p, d_cg_ap.p, d_cg_scal.p, args.tolerance,
                         d_cg_done.p, iter, hist_dev);
        HIP_TRY(hipMemcpyAsync(ctl, d_cg_scal.p, (size_t)(16 + iter) * sizeof(double), hipMemcpyDeviceToHost, stream));
        sync();
        done = ctl[8] != 0.0;
        batch = 4;
      }
    } else {
      // one launch per iteration (cg_step_kernel): start + matvec 1, then update(k - 1) + matvec(k) ..., and the last
      // iteration's update alone before the read-back; same arithmetic, same bits as the two-launch form
      auto step = [&](int it, int mode) {
        launch_cg_step(stream, ne, d_A.p, d_b, d_eleallq, d_cg_res.p, d_cg_p.p, d_cg_ap.p, d_cg_scal.p, args.tolerance,
                       d_cg_done.p, it, hist_dev, mode, results_by_copy ? nullptr : ctl, 16 + it + 1);
      };
      step(1, 1);                                            // iter = the iteration whose matvec is in flight
      while (true) {
        const int last = std::min(args.maxiter - 1, iter + batch - 1);
        for (; iter < last; ++iter) step(iter + 1, 2);
        step(iter, 4);                                        // (stores the control block into `ctl` itself)
        if (results_by_copy) HIP_TRY(hipMemcpyAsync(ctl, d_cg_scal.p, (size_t)(16 + iter + 1) * sizeof(double), hipMemcpyDeviceToHost, stream));
        if (spec_dq) scatter_device(spec_dq, spec_pot, false);      // update_direct: speculative charge write
        sync();
        done = ctl[8] != 0.0;
        if (spec_dq && done) spec_done = true;
        if (done || iter + 1 >= args.maxiter) break;
        ++iter;
        step(iter, 3);
        batch = 4;
      }
    }
    prof.end(stream);
    cg_iterations = done ? (int)ctl[6] : 0;
    if (done) cg_batch = cg_iterations;                      // iteration 1..n: n launches reach convergence next time
    // the log lines of :919-928: one per iteration, the converged one with the net charge
    const int last = done ? cg_iterations : args.maxiter - 1;
    const double *hist = ctl + 16;
    for (int it = 1; it <= last; ++it) {
      if (done && it == last) logf("***** Converged at iteration %d. res = %g netcharge = %g\n", it, hist[it], ctl[7]);
      else logf("Iteration %d: res = %g\n", it, hist[it]);
    }
  }

  // fix_conp.cpp:698-718 equation_solve
  void equation_solve() {
    if (args.minimizer == CONP_SOLVER_CG) cg();
    else inv();
  }

  // fix_conp.cpp:1071-1116 get_setq
  void get_setq(const conp_atoms *at) {
    const int ne = idx.elenum_all;
    if (args.minimizer == CONP_SOLVER_CG) {
      HIP_TRY(hipMemcpyAsync(d_elesetq.p, d_eleallq, ne * sizeof(double), hipMemcpyDeviceToDevice, stream));
    } else {
      launch_gemv_rows(stream, ne, 0, ne, d_A.p, d_b, d_elesetq.p);
    }
    launch_left_sum(stream, ne, d_elecheck.p, d_elesetq.p, d_scalars.p + 0);
    HIP_TRY(hipMemcpyAsync(&totsetq, d_scalars.p + 0, sizeof(double), hipMemcpyDeviceToHost, stream));
    if (args.qinit) {
      std::vector<double> qi(ne_pad, 0.0), mine(std::max(idx.elenum, 1));
      for (int iloc = 0; iloc < idx.elenum; ++iloc) mine[iloc] = at->q[idx.tag2local[idx.ele2tag[iloc]]];
      ele_comm(mine.data(), 1, qi.data());
      d_eleinitq.upload(qi, stream);
    }
    sync();
    if (env.one_electrode) inv_project();
  }

  // fix_conp.cpp:426-464 linalg_setup
  void linalg_setup(const conp_atoms *at) {
    if (runstage != 0) return;
    static const bool tt = getenv("CONP_TIME_REN") != nullptr;         // same switch as the re-neighbour breakdown
    auto t0 = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
      if (!tt) return;
      sync();
      const auto t = std::chrono::steady_clock::now();
      std::fprintf(stderr, "  linalg_setup %-18s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t0).count());
      t0 = t;
    };
    if (args.a_matrix_f == 0) a_cal(at);
    else if (!matrix_loaded) a_read_file(at, args.a_matrix_file);      // fix_conp.cpp:443-446
    mark("a_cal");
    b_setq_cal(at);
    mark("b_setq_cal");
    equation_solve();
    mark("equation_solve");
    get_setq(at);
    mark("get_setq");
    mesgf("conp output: <d,d> = %.8g\n", -totsetq);                      // :458-461
  }

  // ---- per-step device path -------------------------------------------------------------------
  // km_ewald.cpp:153-167 b_cal + fix_conp.cpp:1281-1365 blist_coul_cal, this rank's shard, into d_b
  // dx / dq: this rank's atom arrays (real-space rows, charge write).  The structure factors read the charged electrolyte atoms
  // through an index list: the owned ones out of dx / dq, or -- decomposed runs -- every rank's, gathered into d_xg / d_qg.
  void b_cal_device(const double *dx, const double *dq, bool coulyes, bool timed = false) {
    const int ne = idx.elenum_all;
    pp_elyte_valid = pp_u_valid = false;          // positions / charges may have changed: the mesh caches are this update's or nobody's
    const double *ex = decomposed ? d_xg.p : dx, *eq = decomposed ? d_qg.p : dq;
    const int *eidx = decomposed ? d_iota.p : d_elyte_idx.p;
    if (!kspace_ready || !tables_current || d_b == nullptr)
      throw ConpError(CONP_ERR_STATE, "b_cal before the k tables / electrode phase tables exist (setup_post_neighbor, a_cal)");
    if (timed) {
      for (auto &e : ev_b) if (!e) HIP_TRY(hipEventCreate(&e));
      HIP_TRY(hipEventRecord(ev_b[0], stream));
    }
    const int slab = (kt.slabflag && env.rank == 0) ? 1 : 0;
    // real-space rows: with replicated atoms this rank's row range; a sub-domain's list holds its own electrode atoms' rows only
    const int rr0 = !coulyes ? 0 : (decomposed ? 0 : row0), rr1 = !coulyes ? 0 : (decomposed ? ne : row1);
    bool ride = false, use_fin = false, ride_hc = false;
    BRowArgs fin{}, pairs_keep{};
    if (args.pppm) {
      // `pppm` keyword: the k-space b comes from the mesh (pppm_conp.cpp:269-316); the mesh is not sharded -- rank 0 owns it
      prof.begin("pppm_b", stream);
      // one rank: the real-space pair sums ride in the spread launch and the stencil gather completes the rows of b (slab term,
      // pair sums): no b_real_combine launch behind the mesh (round 4; CONP_NO_RIDE / CONP_NO_FUSE / CONP_TIME_SPLIT: as before)
      const bool pp_fin = env.nranks == 1 && !nccl && !decomposed && !no_fuse && !no_ride && !(timed && time_split);
      BRowArgs pairs = make_brow(ne, ne_pad, rr0, rr1, d_b_rowptr.p, d_b_ele.p, d_b_oth.p, dx, dq, d_type.p, real_params(), 0,
                                 nullptr, 0, nullptr, nullptr, 0, 0.0, nullptr, nullptr);
      fin = make_brow(ne, ne_pad, rr0, rr1, d_b_rowptr.p, d_b_ele.p, d_b_oth.p, dx, dq, d_type.p, real_params(), 1, d_bk.p, slab,
                      d_ele_z.p, d_slab_part.p, 0, 4.0 * 3.14159265358979323846 / kt.volume, d_b, d_scalars.p + 2);
      fin.breal = d_breal.p;
      if (env.rank == 0)
        launch_pppm_b(stream, dpppm, nl, eidx, ex, eq, ne, ne_pad, d_pp_egrid.p, d_pp_ew.p, d_pp_re.p, d_pp_im.p,
                      d_slab_part.p, &n_slab_part, d_bk.p, &pp_im_clean, pp_fin ? &pairs : nullptr, pp_fin ? d_breal.p : nullptr,
                      pp_fin ? &fin : nullptr, pp_keep ? (d_pp_elyte.reserve((size_t)dpppm.nfft), d_pp_elyte.p) : nullptr);
      else
        d_bk.zero(stream);
      ++pp_elyte_spreads;
      pp_elyte_valid = pp_keep && env.rank == 0;
      prof.end(stream);
      use_fin = pp_fin;
    } else {
      // the real-space pair sums depend on x, q only: they ride along in the phase kernel's launch (spare blocks) unless the
      // host-buffer hooks were asked to time the two halves of b_cal separately (CONP_TIME_SPLIT; Ktime / Ctime, fix_conp.cpp:553-568)
      BRowArgs pairs = make_brow(ne, ne_pad, rr0, rr1, d_b_rowptr.p, d_b_ele.p, d_b_oth.p, dx, dq, d_type.p, real_params(), 0,
                                 nullptr, 0, nullptr, nullptr, 0, 0.0, nullptr, nullptr);
      ride = !no_fuse && !(timed && time_split) && !no_ride;
      // small systems on one rank: no phase launch at all -- every sk_gemm segment computes the phase tables of its own atoms in
      // front of its chunk loop, the pair sums ride in spare workgroups of that launch (SkFuse; CONP_NO_PHASE_FUSE: comparison switch).
      // A workgroup of this launch owns its CU (registers): the pair workgroups run on the CUs sk_gemm leaves free, in at most two
      // rounds of ~5 us -- shorter than the ~13 us the sk_gemm workgroups take
      const int nwg_sk = (int)seg_ptr_h.size() - 1;
      const bool fuse_phase = ride && env.nranks == 1 && !nccl && !decomposed && nl_pad <= 4096 && max_seg_chunks <= 4 && n_slab_slots > 0 &&
                              nwg_sk < num_cus && (ne + 7) / 8 <= 2 * (num_cus - nwg_sk) && !no_phase_fuse;
      SkFuse fz;
      std::memset(&fz, 0, sizeof fz);           // (padding bytes too: the block is compared bytewise below)
      if (fuse_phase) {
        fz.on = 1; fz.nl = nl; fz.kzt = plan.kzt; fz.nwg_sk = nwg_sk; fz.elyte_idx = eidx; fz.x = ex; fz.q = eq;
        fz.ux = kt.unitk[0]; fz.uy = kt.unitk[1]; fz.uz = kt.unitk[2];
        fz.Xt = d_Xt.p; fz.Yt = d_Yt.p; fz.Zs = d_Zt.p; fz.qc = d_qc.p; fz.slab_part = d_slab_part.p;
        fz.rows = pairs; fz.breal_out = d_breal.p;
        n_slab_part = n_slab_slots;
        // the kernel reads the block from device memory: uploaded when its content changes (the host's x, q pointers are the same
        // from step to step; a re-neighbour or other arrays change them)
        if (d_skfuse.n == 0 || std::memcmp(&fz, &skfuse_h, sizeof fz) != 0) {
          std::memcpy(&skfuse_h, &fz, sizeof fz);
          d_skfuse.upload(&skfuse_h, 1, stream);
        }
      } else {
        // When the pieces are added by a launch of their own (hc_sum: many pieces per row tile, the headline and slab plans) the
        // pair sums ride THERE, not in the phase launch: that launch sits at the ~5 us floor of any small kernel with a handful of
        // workgroups, the pair rows fill the rest of the chip beside it (17.7 -> 19.2 us for hc_sum + dot), while in the phase
        // kernel they compete with 37 MB of table writes (15.1 -> 10.7 us without them): 0.2791 / 0.2796 -> 0.2763 / 0.2757 ms per
        // update at the headline size, 0.7261 / 0.7268 -> 0.7228 / 0.7244 in the slab geometry (one box, tools/ab_env.sh).
        // CONP_RIDE_PHASE: comparison switch, the phase launch as before.
        ride_hc = ride && diag_switch("CONP_RIDE_PHASE") == nullptr && sk_projects() && n_frags > 0 && nzc > 0 &&
                  zc_final_fits((int)own_rt_h.size(), nzc) &&
                  (zn_use() || !bands_aligned || (hc_presum_env ? atoi(hc_presum_env) != 0 : hslots > 32 * (int)own_rt_h.size()));
        prof.begin("elyte_phase", stream);
        ZnWindow zw{};
        const bool zn_now = zn_use() || zn_gen_use();
        if (zn_now) zw = ZnWindow{d_zn_Bt.p, d_zn_g0c.p, zn_flag_dev, 16 * zn_ncf, zn_n, ZN_W, zn_beta, zn_gscale};
        launch_elyte_phase(stream, nl, nl_pad, eidx, ex, eq, kt.unitk[0], kt.unitk[1], kt.unitk[2], plan.kxmax,
                           plan.kymax, plan.nz, plan.kzt, 1 + plan.n_col_tiles * 32, d_Xt.p, d_Yt.p, d_Zt.p, d_qc.p, d_slab_part.p,
                           &n_slab_part, ride && !ride_hc ? &pairs : nullptr, d_breal.p, 16 * table_c0, 16 * table_c1, zn_now ? &zw : nullptr);
        zs_current = !zn_now; last_ex = ex; last_eq = eq; last_eidx = eidx;
        prof.end(stream);
        pairs_keep = pairs;
      }
      // with the pair sums in hand and a small z-class table the dot kernel can finish b itself: no b_real_combine launch
      fin = make_brow(ne, ne_pad, rr0, rr1, d_b_rowptr.p, d_b_ele.p, d_b_oth.p, dx, dq, d_type.p, real_params(), 1, d_bk.p, slab,
                      d_ele_z.p, d_slab_part.p, n_slab_part, 4.0 * 3.14159265358979323846 / kt.volume, d_b, d_scalars.p + 2);
      fin.breal = d_breal.p;
      use_fin = ride && nzc > 0 && zc_final_fits((int)own_rt_h.size(), nzc);
      const bool proj = sk_projects();
      if (zn_use() && !fuse_phase) {
        // the z-window form (conp_zn.hip): window matrix of the update, the contraction with 32 / 48 columns, the ranges' pieces
        zn_ensure_tables();
        prof.begin("zn_gemm", stream);
        launch_zn_gemm(stream, dplan, zn_ncf, d_zn_items.p, (int)zn_items_h.size(), d_Xt.p, d_Yt.p, d_zn_Bt.p, d_zn_P.p, zn_n, nzc,
                       d_zn_pieces.p, sk_hc_stride());
        prof.end(stream);
        g_current = false;
        prof.begin("reduce_project", stream);
        launch_project_zclass_pieces(stream, dplan, ne_pad, (int)own_rt_h.size(), d_own_rt.p, nzc, d_zn_pieces.p, d_hslot_ptr.p, d_hslot_idx.p,
                                     true, d_zn_frag_ptr.p, d_zn_frag_ents.p, zn_nfrag, d_Rp.p, d_Xe.p, d_Ye.p, d_own_pv.p, d_zclass.p, d_Hc.p,
                                     d_bk.p, use_fin ? &fin : nullptr, ride_hc ? &pairs_keep : nullptr, d_breal.p, nullptr, 1u << 16,
                                     zn_nrg > 32 /*a piece per range: 32 threads per element*/);
        prof.end(stream);
      } else if (zn_gen_use() && !fuse_phase) {
        // rough electrodes: the ranges' windows -> the z grid -> G and w o G (what sk_reduce leaves), then the general projection
        zn_ensure_tables();
        prof.begin("zn_gemm", stream);
        launch_zn_gemm(stream, dplan, zn_ncf, d_zn_items.p, (int)zn_items_h.size(), d_Xt.p, d_Yt.p, d_zn_Bt.p, nullptr, zn_n, 0,
                       d_zn_pieces.p, 0);
        prof.end(stream);
        prof.begin("sk_reduce", stream);
        launch_zn_windows_to_g(stream, dplan, plan.kzt, zn_n, (int)own_rt_h.size(), d_own_rt.p, zn_nrg, 16 * zn_ncf, d_zn_cov_ptr.p, d_zn_cov_ent.p,
                               d_zn_pieces.p, d_zn_grid.p, d_zn_Dt.p, d_G.p, d_Gw.p);
        prof.end(stream);
        g_current = true;
        prof.begin("b_project", stream);
        launch_b_project(stream, dplan, ne_pad, d_ct_ptr.p, d_tiles.p, d_Gw.p, d_Rp.p, d_Tz.p, d_bk.p);
        prof.end(stream);
      } else {
      reserve_partials();
      prof.begin("sk_gemm", stream);
      launch_sk_gemm(stream, dplan, d_witems.p, witems_maxseg, nwg_sk, nl_pad, d_Xt.p, d_Yt.p, d_Zt.p, d_qc.p,
                     proj ? d_Hpart.p : d_Gpart.p, proj ? d_skproj.p : nullptr, fuse_phase ? d_skfuse.p : nullptr, ne,
                     reinterpret_cast<unsigned *>(d_scalars.p + 8));
      prof.end(stream);
      g_current = !proj;
      if (proj) {
        // planar electrodes: the segments left their projected pieces (128 x nzc each); the dot kernel adds them per row tile
        // (a launch of its own first when there are many: every block of the dot kernel would re-add them all)
        const bool presum = !bands_aligned || (hc_presum_env ? atoi(hc_presum_env) != 0 : hslots > 32 * (int)own_rt_h.size());
        prof.begin("reduce_project", stream);
        launch_project_zclass_pieces(stream, dplan, ne_pad, (int)own_rt_h.size(), d_own_rt.p, nzc, d_Hpart.p, d_hslot_ptr.p, d_hslot_idx.p,
                                     presum, d_frag_ptr.p, d_frag_ents.p, n_frags, d_Rp.p, d_Xe.p, d_Ye.p, d_own_pv.p, d_zclass.p, d_Hc.p, d_bk.p,
                                     use_fin ? &fin : nullptr, ride_hc ? &pairs_keep : nullptr, d_breal.p,
                                     (hc_fused && !(timed && time_split)) ? reinterpret_cast<unsigned *>(d_scalars.p + 8) : nullptr, hc_spin_limit);
        prof.end(stream);
      } else if (nzc > 0 && plan.n_col_tiles == 1 && !no_fuse) {
        // planar electrodes, one column tile: partial-tile sum + Hc product fused, then the per-atom dot (+ row assembly)
        prof.begin("reduce_project", stream);
        launch_reduce_project_zclass(stream, dplan, d_tiles.p, (int)tiles_h.size(), max_nsplit, d_Gpart.p, d_G.p, ne_pad,
                                     (int)own_rt_h.size(), d_own_rt.p, nzc, d_Tzc.p, d_Rp.p, d_Xe.p, d_Ye.p, d_own_pv.p, d_zclass.p,
                                     d_Hc.p, d_bk.p, use_fin ? &fin : nullptr);
        prof.end(stream);
      } else {
        prof.begin("sk_reduce", stream);
        launch_sk_reduce(stream, dplan, d_tiles.p, (int)tiles_h.size(), max_nsplit, d_Gpart.p, d_G.p, d_Gw.p);
        prof.end(stream);
        prof.begin("b_project", stream);
        if (nzc > 0)
          launch_b_project_zclass(stream, dplan, ne_pad, d_rt_mine.p, (int)own_rt_h.size(), d_own_rt.p, nzc, d_Gw.p, d_Tzc.p, d_Rp.p,
                                  d_Xe.p, d_Ye.p, d_own_pv.p, d_zclass.p, d_Hc.p, d_bk.p, use_fin ? &fin : nullptr);
        else
          launch_b_project(stream, dplan, ne_pad, d_ct_ptr.p, d_tiles.p, d_Gw.p, d_Rp.p, d_Tz.p, d_bk.p);
        prof.end(stream);
      }
      }
    }
    if (timed) HIP_TRY(hipEventRecord(ev_b[1], stream));
    if (!use_fin) {
      prof.begin("b_real_combine", stream);
      launch_b_real_combine(stream, ne, ne_pad, rr0, rr1, d_b_rowptr.p, d_b_ele.p, d_b_oth.p, dx, dq,
                            d_type.p, real_params(), 1, d_bk.p, slab, d_ele_z.p, d_slab_part.p, n_slab_part,
                            4.0 * 3.14159265358979323846 / kt.volume, d_b, d_scalars.p + 2, ride ? d_breal.p : nullptr);
      prof.end(stream);
    }
    if (timed) { HIP_TRY(hipEventRecord(ev_b[2], stream)); ev_pending = true; }
    HIP_TRY(hipGetLastError());   // a refused launch (bad grid / LDS size) must not pass silently
  }

  // Btime / Ctime / Ktime of fix_conp.cpp:549-552 and km/blist timers, from the events around the two halves of b_cal
  void collect_b_times() {
    if (!ev_pending) return;
    ev_pending = false;
    float k_ms = 0.f, c_ms = 0.f;
    HIP_TRY(hipEventSynchronize(ev_b[2]));
    HIP_TRY(hipEventElapsedTime(&k_ms, ev_b[0], ev_b[1]));
    HIP_TRY(hipEventElapsedTime(&c_ms, ev_b[1], ev_b[2]));
    Ktime += 1e-3 * k_ms; Ctime += 1e-3 * c_ms; Btime += 1e-3 * (k_ms + c_ms);
  }

  void mesgf(const char *fmt, ...) {
    char line[256];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    mesgbuf += line;
  }

  void logf(const char *fmt, ...) {
    char line[256];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    if (logbuf.size() > (1u << 20)) logbuf.clear();      // a host that never drains (device-resident loops) must not grow it forever
    logbuf += line;
  }

  // the matrix the GEMV streams: the full projected inverse, or -- several ranks -- this rank's rows of it (row r at base + r * Ne)
  const double *s_base() const { return s_sharded ? d_Srows.p - (size_t)row0 * idx.elenum_all : d_A.p; }
  // several ranks: after the setup only the rows [row0, row1) of S are ever read (fix_conp.cpp:1135-1139 does the same row-local
  // ddot_).  Keep those, release the Ne x Ne buffer: 2.1 GB -> 268 MB per rank at Ne = 16384 on 8 ranks.
  void shard_rows() {
    if (s_sharded || !setup_sharded() || args.minimizer != CONP_SOLVER_INV || runstage < 3) return;
    const size_t ne = idx.elenum_all, nr = (size_t)(row1 - row0);
    d_Srows.reserve(std::max<size_t>(nr * ne, 1));
    if (nr) HIP_TRY(hipMemcpyAsync(d_Srows.p, d_A.p + (size_t)row0 * ne, nr * ne * sizeof(double), hipMemcpyDeviceToDevice, stream));
    sync();
    d_A.release();
    s_sharded = true;
  }
  // the whole matrix back in d_A (read-back entry points, `matout`): all-gather of the row blocks
  void unshard_rows() {
    if (!s_sharded) return;
    const size_t ne = idx.elenum_all, nr = (size_t)(row1 - row0);
    d_A.reserve(std::max<size_t>((size_t)rows_per * env.nranks, ne) * ne);
    if (nccl) {
      if (nr) HIP_TRY(hipMemcpyAsync(d_A.p + (size_t)env.rank * rows_per * ne, d_Srows.p, nr * ne * sizeof(double), hipMemcpyDeviceToDevice, stream));
      g_rccl.ok(g_rccl.AllGather(d_A.p + (size_t)env.rank * rows_per * ne, d_A.p, (size_t)rows_per * ne, ncclDouble, nccl, stream), "all-gather(S)");
    } else {
      std::vector<double> mine(std::max<size_t>(nr * ne, 1)), all(ne * ne);
      if (nr) HIP_TRY(hipMemcpyAsync(mine.data(), d_Srows.p, nr * ne * sizeof(double), hipMemcpyDeviceToHost, stream));
      sync();
      std::vector<int> rows(env.nranks);
      for (int r = 0; r < env.nranks; ++r) rows[r] = std::min<int>((int)ne, (r + 1) * rows_per) - std::min<int>((int)ne, r * rows_per);
      rc.gatherv(mine.data(), rows, (int)ne, all.data());
      HIP_TRY(hipMemcpyAsync(d_A.p, all.data(), ne * ne * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    sync();
    d_Srows.release();
    s_sharded = false;
  }

  // fix_conp.cpp:1135-1139: rows [row0,row1) of eleallq = S b (inverse solver), or the CG solve
  void solve_device() {
    const int ne = idx.elenum_all;
    if (args.minimizer == CONP_SOLVER_INV) {
      if (runstage < 3) throw ConpError(CONP_ERR_STATE, "solve before the inverse exists");
      if (!s_sharded && setup_sharded()) shard_rows();          // first update after the setup
      prof.begin("gemv", stream);
      launch_gemv_rows(stream, ne, row0, row1, s_base(), d_b, d_eleallq);
      prof.end(stream);
    } else {
      cg();
    }
  }

  // ---- the two exchanges of an update: all-reduce(b), all-gather(q) ---------------------------------------------
  // RCCL on the library's stream when a communicator exists (one rank per GPU), else the host's callbacks through the
  // page-locked staging buffer (ranks that share a GPU, the LAMMPS glue's default).  MPI_Allgatherv + MPI_Allreduce of the
  // reference: fix_conp.cpp:643 (b_comm of b and of q), :1356 (newtonbuf).
  void allreduce_b() {
    if (env.nranks <= 1 && !nccl) return;
    const int ne = idx.elenum_all;
    prof.begin("allreduce_b", stream);
    if (nccl) g_rccl.ok(g_rccl.AllReduce(d_b, d_b, (size_t)ne, ncclDouble, ncclSum, nccl, stream), "all-reduce(b)");
    else if (rc.active()) {
      double *h = pinned((size_t)ne_pad + 8);
      HIP_TRY(hipMemcpyAsync(h, d_b, ne * sizeof(double), hipMemcpyDeviceToHost, stream));
      sync();
      rc.sum(h, ne);
      HIP_TRY(hipMemcpyAsync(d_b, h, ne * sizeof(double), hipMemcpyHostToDevice, stream));
#ifdef CONP_DIAG
    } else if (getenv("CONP_RANK_EMULATION")) {
      // diagnostic library only (tools/rank_emulation.py): one rank's compute time measured on a box that has no partner ranks
#endif
    } else throw ConpError(CONP_ERR_STATE, "an update on several ranks needs conp_fix_comm_init_rccl or conp_fix_set_comm (or do the "
                                           "two collectives yourself between conp_fix_b_cal_device / _solve_device / _scatter_device)");
    prof.end(stream);
  }
  void allgather_q() {
    if ((env.nranks <= 1 && !nccl) || args.minimizer != CONP_SOLVER_INV) return;    // CG solves all rows on every rank, like the reference
    const int ne = idx.elenum_all;
    prof.begin("allgather_q", stream);
    if (nccl) {
      double *mine = d_eleallq + (size_t)env.rank * rows_per;
      g_rccl.ok(g_rccl.AllGather(mine, d_eleallq, (size_t)rows_per, ncclDouble, nccl, stream), "all-gather(q)");
    } else if (rc.active()) {
      double *h = pinned((size_t)ne_pad + 8 + (size_t)ne + rows_per) + ne_pad + 8;
      const int nr = row1 - row0;
      if (nr) HIP_TRY(hipMemcpyAsync(h + ne, d_eleallq + row0, nr * sizeof(double), hipMemcpyDeviceToHost, stream));
      sync();
      std::vector<int> rows(env.nranks);
      for (int r = 0; r < env.nranks; ++r) rows[r] = std::min(ne, (r + 1) * rows_per) - std::min(ne, r * rows_per);
      rc.gatherv(h + ne, rows, 1, h);
      HIP_TRY(hipMemcpyAsync(d_eleallq, h, ne * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    prof.end(stream);
  }

  // plain `fix conp` on one rank with the inverse solver: GEMV and charge write in one launch (gemv_finish_kernel)
  bool can_fuse_solve() const {
    return !no_fuse && args.minimizer == CONP_SOLVER_INV && !args.conq && !args.cond && env.nranks == 1 && !nccl && !s_sharded &&
           runstage >= 3;
  }
  // The projected inverse as a symmetric matrix (conp_kernels.hip "GEMV ... as a SYMMETRIC matrix"): from 2048 electrode atoms up
  // the fused solve reads packed lower-triangle tiles, half the bytes of the row-by-row product.  Packed once per matrix
  // (spk_of: the matrix generation it was made from).  CONP_GEMV_FULL: comparison switch, always the row-by-row product.
  long s_generation = 0, spk_of = -1;
  bool spk_symmetric = false;       // the matrix generation spk_of passed the symmetry test of the packing pass
  DevBuf<unsigned long long> d_symstat;
  bool use_sym_gemv() const {
    return !gemv_rows && idx.elenum_all >= 2048;
  }
  void solve_scatter_fused(double *d_q_atoms, double potdiff) {
    const int ne = idx.elenum_all;
    if (use_sym_gemv()) {
      if (spk_of != s_generation) {
        d_Spk.reserve(sym_packed_doubles(ne_pad));
        d_yp.reserve((size_t)(ne_pad / 128) * ne_pad);
        d_symstat.reserve(2);
        launch_sym_pack(stream, ne, ne_pad, d_A.p, d_Spk.p, d_symstat.p);
        // once per matrix: is it symmetric?  The library's own projected inverse is (to ~1e-16 of its largest entry); a matrix
        // from a file or from conp_fix_set_matrix need not be, and the reference multiplies full rows -- such a matrix keeps the
        // row-by-row product below instead of being symmetrised without a word.
        unsigned long long st[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(st, d_symstat.p, sizeof st, hipMemcpyDeviceToHost, stream));
        sync();
        double mx, md;
        std::memcpy(&mx, &st[0], sizeof mx); std::memcpy(&md, &st[1], sizeof md);
        spk_symmetric = md <= 1e-10 * mx;
        spk_of = s_generation;
        if (!spk_symmetric)
          mesgf("conp/hip: the solve matrix is not symmetric (max |S_ij - S_ji| = %.3g, max |S_ij| = %.3g): full rows are multiplied\n", md, mx);
      }
    }
    if (use_sym_gemv() && spk_symmetric) {
      prof.begin("gemv_charge", stream);
      launch_sym_gemv_finish(stream, ne, ne_pad, d_Spk.p, d_b, d_yp.p, d_eleallq, d_elesetq.p, args.qinit ? d_eleinitq.p : nullptr,
                             potdiff, d_ele_csr_ptr.p, d_ele_csr_of.p, d_ele_csr_row.p, d_qele.p, d_q_atoms);
      prof.end(stream);
      left_stale = true; left_potdiff = potdiff;
      HIP_TRY(hipGetLastError());
      return;
    }
    prof.begin("gemv_charge", stream);
    launch_gemv_finish(stream, ne, d_A.p, d_b, d_eleallq, d_elesetq.p, args.qinit ? d_eleinitq.p : nullptr, potdiff, d_ele_csr_ptr.p,
                       d_ele_csr_of.p, d_qele.p, d_q_atoms);
    prof.end(stream);
    left_stale = true; left_potdiff = potdiff;        // the fix scalar's group-1 sum is formed on demand (refresh_scalar)
    HIP_TRY(hipGetLastError());
  }

  // fix_conp.cpp:1149-1159: charges for owned + ghost electrode atoms, scalar output
  void scatter_device(double *d_q_atoms, double potdiff, bool labelled = true) {
    const int ne = idx.elenum_all;
    if (labelled) prof.begin("charge_write", stream);
    if (args.cond) {
      // fix cond (fix_cond.cpp:58-126): vmult once (cond_setup2), then the potential from the cell dipole
      if (!cond_ready) {
        std::vector<double> sq(ne), sz(ne);
        HIP_TRY(hipMemcpyAsync(sq.data(), d_elesetq.p, ne * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(sz.data(), d_setzvec.p, ne * sizeof(double), hipMemcpyDeviceToHost, stream));
        sync();
        double zOAz = 0.;
        for (int i = 0; i < ne; ++i) zOAz += sq[i] * sz[i];
        double vm = 4 * 3.14159265358979323846 * zOAz * env.zprd / (evscale * env.xprd * env.yprd);
        vm /= 1 + vm;
        vm /= zOAz;
        cond_vmult = vm;
        cond_ready = true;
      }
      launch_cond_potdiff(stream, ne, d_setzvec.p, d_eleallq, d_slab_part.p, n_slab_part, env.zprd, potdiff, cond_vmult,
                          d_scalars.p + 3);
      launch_charge_finish(stream, ne, n_ele_atoms, d_ele_pairs.p, d_elecheck.p, d_eleallq, d_elesetq.p,
                           args.qinit ? d_eleinitq.p : nullptr, 0.0, d_scalars.p + 3, d_qele.p, d_q_atoms, nullptr);
    } else if (args.conq) {
      // fix conq (fix_conq.cpp:41-90): `potdiff` carries the prescribed charge QR; the potential difference follows from
      // the group-1 sum of S b, all on the device
      launch_left_sum(stream, ne, d_elecheck.p, d_eleallq, d_scalars.p + 1);
      launch_conq_potdiff(stream, d_scalars.p + 1, potdiff, totsetq, env.one_electrode, d_scalars.p + 3);
      launch_charge_finish(stream, ne, n_ele_atoms, d_ele_pairs.p, d_elecheck.p, d_eleallq, d_elesetq.p,
                           args.qinit ? d_eleinitq.p : nullptr, 0.0, d_scalars.p + 3, d_qele.p, d_q_atoms, nullptr);
    } else {
      launch_charge_finish(stream, ne, n_ele_atoms, d_ele_pairs.p, d_elecheck.p, d_eleallq, d_elesetq.p,
                           args.qinit ? d_eleinitq.p : nullptr, potdiff, nullptr, d_qele.p, d_q_atoms, d_scalars.p + 1);
    }
    if (labelled) prof.end(stream);
    HIP_TRY(hipGetLastError());
  }

  void finish_scalar(double potdiff) {
    if (left_stale) { launch_left_sum(stream, idx.elenum_all, d_elecheck.p, d_eleallq, d_scalars.p + 1); left_stale = false; }
    double *h = pinned((size_t)ne_pad + 8) + ne_pad;
    HIP_TRY(hipMemcpyAsync(h, d_scalars.p, 4 * sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    scalar_output = (args.conq || args.cond) ? h[3] : potdiff * totsetq + h[1];   // fix_conp.cpp:1159 / fix_conq.cpp:78-80 / fix_cond.cpp:116
    slabcorr = h[2];
  }

  // fix_conp.cpp:1120-1161 update_charge (host-buffer flavour)
  void update_charge(const conp_atoms *at, double potdiff) {
    const int ne = idx.elenum_all;
    if (can_fuse_solve()) solve_scatter_fused(d_q.p, potdiff);
    else {
      if (args.minimizer == CONP_SOLVER_INV) { solve_device(); allgather_q(); }
      scatter_device(d_q.p, potdiff);        // the device copy of q follows atom->q (post_force of the same step reuses it)
    }
    double t0 = time_host ? now_s() : 0.0;
    double *qe = pinned((size_t)ne_pad + 8);
    if (results_by_copy) {
      HIP_TRY(hipMemcpyAsync(qe, d_qele.p, ne * sizeof(double), hipMemcpyDeviceToHost, stream));
      finish_scalar(potdiff);               // one synchronisation for the charges and the scalars
    } else {
      // charges and scalars stored into the page-locked staging area by ONE kernel (which also forms the fix scalar's group-1 sum):
      // no copy-engine transfer at the end of the update
      double *h = qe + ne_pad;
      launch_results_out(stream, ne, d_elecheck.p, d_eleallq, d_scalars.p, left_stale, d_qele.p, qe, h);
      left_stale = false;
      sync();
      scalar_output = (args.conq || args.cond) ? h[3] : potdiff * totsetq + h[1];
      slabcorr = h[2];
    }
    collect_b_times();
    if (time_host) { const double t1 = now_s(); th[3] += t1 - t0; t0 = t1; }
    // owned and ghost electrode atoms :1153-1158, through the (atom, row) list of the last post_neighbor (the atom arrays keep
    // their order between re-neighbourings; b_cal checked the count) instead of a walk over all atoms
    for (int k = 0; k < n_ele_atoms; ++k) at->q[ele_pairs_h[2 * (size_t)k]] = qe[ele_pairs_h[2 * (size_t)k + 1]];
    if (time_host) { th[4] += now_s() - t0; th[5] += 1.0; }
  }

  // fix_conp.cpp:577-580 post_force -> :1163-1201 force_cal + :1368-1444 blist_coul_cal_post_force
  // `ntimestep` >= 0: the host promises that x is what it handed to pre_force at that step (LAMMPS: nothing moves between
  // pre_force and post_force of a step); the device copy -- positions, and charges incl. the new electrode charges -- is then
  // reused instead of uploaded again.  Forces and the accumulators come back through the page-locked staging buffer.
  bool pf_f_dirty = true;              // the device force array may hold non-zeros (first use, or the last call had contributions)
  int64_t resident_step = -1;          // step whose x, q are on the device (host pre_force), -1: none
  const void *resident_x = nullptr;
  void post_force(const conp_atoms *at, double *f, double *ek, double *ec, double *vir, int64_t ntimestep = -1) {
    if (at->nlocal + at->nghost != nall) throw ConpError(CONP_ERR_STATE, "atom count changed without post_neighbor");
    const bool resident = ntimestep >= 0 && ntimestep == resident_step && at->x == resident_x;
    if (!resident) upload_xq(at);   // charges were just updated by pre_force on the host side
    const size_t nf = (size_t)nall * 3;
    if (d_f.n < nf) pf_f_dirty = true;             // fresh allocation: contents undefined
    d_f.reserve(nf); d_pfacc.reserve(9);
    prof.begin("post_force", stream);
    launch_post_force(stream, bl_inum, d_bl_ilist.p, d_bl_numneigh.p, d_bl_first.p, d_bl_neigh.p, at->nlocal, nall, env.newton_pair != 0, d_x.p, d_q.p, d_type.p,
                      d_atom2eleall.p, real_params(), env.qqrd2e, d_f.p, d_pfacc.p, pf_f_dirty);
    prof.end(stream);
    double *acc = pinned((size_t)ne_pad + 8 + nf + 16) + ne_pad + 8 + nf;
    HIP_TRY(hipMemcpyAsync(acc, d_pfacc.p, 9 * sizeof(double), hipMemcpyDeviceToHost, stream));
    sync();
    // the correction only acts where the Gaussians overlap (eta^2 r^2 < 5.8, r < 1.2 A at eta = 1.979): in a normal MD step no
    // pair is that close, the force array on the device is still all zero and is not brought over
    pf_f_dirty = acc[8] > 0.0;
    if (decomposed) rc.sum(acc + 7, 1);            // MPI_Allreduce of the electrode sum q^2 (fix_conp.cpp:1176, :1193)
    if (pf_f_dirty) {
      double *fh = acc - nf;
      HIP_TRY(hipMemcpyAsync(fh, d_f.p, nf * sizeof(double), hipMemcpyDeviceToHost, stream));
      sync();
      if (f) for (size_t k = 0; k < nf; ++k) f[k] += fh[k];
    }
    if (ek) *ek = ehgo_active ? env.qqrd2e * 1.0 * acc[7]                                                    // :1198
                              : env.qqrd2e * 1.0 * args.eta * acc[7] / (std::sqrt(2.0) * 1.77245385090551602729);   // :1180
    if (ec) *ec = acc[0];
    if (vir) for (int k = 0; k < 6; ++k) vir[k] = acc[1 + k];
  }

  // fix_conp.cpp:677-695 b_cal / update_bk
  void b_cal(const conp_atoms *at) {
    if (at->nlocal + at->nghost != nall) throw ConpError(CONP_ERR_STATE, "atom count changed without post_neighbor");
    if (!tables_current) km_a_read(at);      // electrode phase tables (kspmod->a_read) not built yet: b_cal before a_cal
    double t0 = time_host ? now_s() : 0.0;
    // the transfer of x, q is started first: the membership walk over the owned atoms below (30 us at the headline size) then runs
    // while the DMA does -- neither depends on the other
    upload_xq(at);
    if (time_host) { const double t1 = now_s(); th[1] += t1 - t0; t0 = t1; }
    if (elyte_list_stale(at)) { sync(); build_elyte_list(at); }       // km_ewald.cpp:686 is evaluated every step
    if (time_host) { const double t1 = now_s(); th[0] += t1 - t0; t0 = t1; }
    if (decomposed) gather_elyte(at);
    b_cal_device(d_x.p, d_q.p, true, true);
    if (time_host) { const double t1 = now_s(); th[2] += t1 - t0; t0 = t1; }
    // (a handle that is rank r of N WITHOUT a communicator leaves this rank's shard in b: the caller sums the shards itself)
    if (nccl || rc.active()) allreduce_b();
  }

  // ---- one device-resident update as a HIP graph ------------------------------------------------
  // The decks' updates are launch-bound (7-8 dependent kernels of 5-15 us each), the textbook case for a graph.  Measured on
  // MI355X / ROCm 7.2 it does NOT pay: graph replay is 5-12 % slower than the plain in-order launches (il_onelayer 57.2 vs
  // 51.2 us, dilute 47.5 vs 44.8, cond2 78.1 vs 71.5, headline 374 vs 371 us per update), so it is opt-in (CONP_GRAPH=1) and
  // kept for re-measurement on later ROCm releases.  The graph is dropped by every C-ABI call that can change buffers, plans
  // or parameters (drop_graph()) and re-captured on the next update; a potential difference that changes from step to step
  // (`v_name`) keeps the direct-launch path.
  hipGraph_t upd_graph = nullptr;
  hipGraphExec_t upd_exec = nullptr;
  const double *g_dx = nullptr;
  double *g_dq = nullptr;
  double g_pot = 0.0;
  int g_warm = 0, g_pot_changes = 0;
  bool graph_off = env_knob("CONP_GRAPH") == nullptr || atoi(env_knob("CONP_GRAPH")) == 0;
  // called at the top of every C-ABI entry that may touch the device: the handle's device becomes the thread's current one
  // (a host that drives several GPUs from one thread may have switched), and a captured update graph is dropped
  void drop_graph() {
    (void)hipSetDevice(env.device);
    if (upd_exec) { (void)hipGraphExecDestroy(upd_exec); upd_exec = nullptr; }
    if (upd_graph) { (void)hipGraphDestroy(upd_graph); upd_graph = nullptr; }
    g_warm = 0;
  }
  void update_direct(const double *dx, double *dq, double potdiff) {
    if (decomposed) throw ConpError(CONP_ERR_STATE, "device-resident updates take replicated atoms (conp_env.rank / nranks); "
                                                    "spatially decomposed runs use the host-buffer hooks");
    if (zn_overflowed())
      throw ConpError(CONP_ERR_NUMERIC, "a device-resident update's z-window overflowed (an electrolyte atom drifted more than 2.5 A in z since "
                                          "the last conp_fix_post_neighbor): the charges of the updates since then are invalid; the full kernels are used from now on");
    b_cal_device(dx, dq, true);
    if (can_fuse_solve()) { solve_scatter_fused(dq, potdiff); return; }
    allreduce_b();
    // CG on one rank: the charge write is enqueued behind every batch of iterations BEFORE the host reads the convergence flag
    // (cg()): the device does not idle through the read-back and a launch from cold; a batch that did not converge just wrote
    // intermediate charges, which the next batch's write replaces
    const bool spec = args.minimizer == CONP_SOLVER_CG && env.nranks <= 1 && !nccl && !args.conq && !args.cond && !cg_no_spec;
    if (spec) { spec_dq = dq; spec_pot = potdiff; spec_done = false; }
    solve_device();
    spec_dq = nullptr;
    allgather_q();
    if (!(spec && spec_done)) scatter_device(dq, potdiff);
  }
  const bool cg_no_spec = diag_switch("CONP_CG_NO_SPEC") != nullptr;      // comparison switch: charge write after the read-back
  double *spec_dq = nullptr;
  double spec_pot = 0.0;
  bool spec_done = false;
  void update_device(const double *dx, double *dq, double potdiff) {
    // (the legacy default stream cannot be captured)
    const bool can = !graph_off && stream != nullptr && !prof.on && args.minimizer == CONP_SOLVER_INV && runstage >= 3 &&
                     (!args.cond || cond_ready) && env.nranks == 1;
    if (can && upd_exec && potdiff != g_pot && ++g_pot_changes > 2) graph_off = true;     // variable potential: stay direct
    if (!can || graph_off) { if (upd_exec) drop_graph(); update_direct(dx, dq, potdiff); return; }
    if (upd_exec && (dx != g_dx || dq != g_dq || potdiff != g_pot)) drop_graph();
    if (!upd_exec) {
      // the first update after a (re)setup runs directly (lazy allocations happen there); the second one is captured
      if (g_warm++ == 0) { update_direct(dx, dq, potdiff); return; }
      if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        graph_off = true;
        update_direct(dx, dq, potdiff);
        return;
      }
      try {
        update_direct(dx, dq, potdiff);
      } catch (...) {
        hipGraph_t junk = nullptr;
        (void)hipStreamEndCapture(stream, &junk);
        if (junk) (void)hipGraphDestroy(junk);
        graph_off = true;
        throw;
      }
      HIP_TRY(hipStreamEndCapture(stream, &upd_graph));
      if (hipGraphInstantiate(&upd_exec, upd_graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        upd_exec = nullptr; drop_graph(); graph_off = true;
        update_direct(dx, dq, potdiff);
        return;
      }
      g_dx = dx; g_dq = dq; g_pot = potdiff;
    }
    HIP_TRY(hipGraphLaunch(upd_exec, stream));
  }

  // fix_conp.cpp:543-573 pre_force
  void pre_force(const conp_atoms *at, int64_t ntimestep, double potdiff) {
    if (runstage < 2) throw ConpError(CONP_ERR_STATE, "pre_force before setup_pre_force");
    if (ntimestep % args.everynum != 0) return;
    resident_step = -1;
    b_cal(at);
    if (args.minimizer == CONP_SOLVER_CG) equation_solve();
    update_charge(at, potdiff);
    if (zn_overflowed()) {               // (update_charge ended with a synchronisation: the flag is this update's)
      b_cal(at);
      if (args.minimizer == CONP_SOLVER_CG) equation_solve();
      update_charge(at, potdiff);
    }
    resident_step = ntimestep; resident_x = at->x;
  }
};

// =================================================================================================
// C ABI
// =================================================================================================
#define CONP_GUARD_BEGIN try {
#define CONP_GUARD_END                                                   \
  }                                                                      \
  catch (const ConpError &e) { g_last_error = e.what(); return e.code; } \
  catch (const std::exception &e) { g_last_error = e.what(); return CONP_ERR_STATE; } \
  return CONP_OK;

extern "C" {

const char *conp_last_error(void) { return g_last_error.c_str(); }
int conp_abi_version(void) { return CONP_ABI_VERSION; }

// fix_conp.cpp:79-176.  arg[0..2] = ID group1 conp ; arg[3] Nevery ; arg[4] group2 ; arg[5] eta ; arg[6] DV ; arg[7] log
int conp_parse_fix_args(int narg, const char *const *arg, int ntypes, conp_fix_args *out) {
  CONP_GUARD_BEGIN
  if (!out) throw ConpError(CONP_ERR_ARG, "null output");
  std::memset(out, 0, sizeof(*out));
  if (narg < 8) throw ConpError(CONP_ERR_ARG, "Illegal fix conp command (too few input parameters)");
  auto numeric = [&](const char *s, const char *what) {
    char *end = nullptr;
    const double v = std::strtod(s, &end);
    if (end == s || *end != '\0') throw ConpError(CONP_ERR_ARG, std::string("Expected floating point parameter instead of '") + s + "' in " + what);
    return v;
  };
  auto inumeric = [&](const char *s, const char *what) {
    char *end = nullptr;
    const long v = std::strtol(s, &end, 10);
    if (end == s || *end != '\0') throw ConpError(CONP_ERR_ARG, std::string("Expected integer parameter instead of '") + s + "' in " + what);
    return (int)v;
  };
  out->maxiter = 100; out->tolerance = 0.000001; out->minimizer = CONP_SOLVER_INV;   // :88-90
  out->lowmem = 1; out->nullneutral = 1; out->ff_flag = CONP_FF_NORMAL;
  out->conq = std::strncmp(arg[2], "conq", 4) == 0;   // FixStyle(conq,FixConq) fix_conq.h:21
  out->cond = std::strncmp(arg[2], "cond", 4) == 0;   // FixStyle(cond,FixCond) fix_cond.h
  out->everynum = inumeric(arg[3], "fix conp Nevery");
  if (out->everynum <= 0) throw ConpError(CONP_ERR_ARG, "Illegal fix conp command (Nevery must be positive)");
  std::snprintf(out->group2, sizeof(out->group2), "%s", arg[4]);
  out->eta = numeric(arg[5], "fix conp eta");
  if (std::strncmp(arg[6], "v_", 2) == 0) {
    out->potdiff_is_variable = 1;
    std::snprintf(out->potdiff_var, sizeof(out->potdiff_var), "%s", arg[6] + 2);
  }
  else out->potdiff = numeric(arg[6], "fix conp DV");
  std::snprintf(out->logfile, sizeof(out->logfile), "%s", arg[7]);
  for (int iarg = 8; iarg < narg; ++iarg) {
    const char *a = arg[iarg];
    if (!std::strcmp(a, "ffield")) {
      if (out->ff_flag == CONP_FF_NOSLAB) throw ConpError(CONP_ERR_ARG, "Invalid fix conp command (ffield and noslab cannot both be chosen)");
      out->ff_flag = CONP_FF_FFIELD;
    } else if (!std::strcmp(a, "noslab")) {
      if (out->ff_flag == CONP_FF_FFIELD) throw ConpError(CONP_ERR_ARG, "Invalid fix conp command (ffield and noslab cannot both be chosen)");
      out->ff_flag = CONP_FF_NOSLAB;
    } else if (!std::strcmp(a, "org") || !std::strcmp(a, "inv")) {
      if (out->a_matrix_f != 0) throw ConpError(CONP_ERR_ARG, "Invalid fix conp command (A matrix file specified more than once)");
      out->a_matrix_f = !std::strcmp(a, "org") ? 1 : 2;
      if (++iarg >= narg) throw ConpError(CONP_ERR_ARG, "Invalid fix conp command (No A matrix filename given)");
      std::snprintf(out->a_matrix_file, sizeof(out->a_matrix_file), "%s", arg[iarg]);
    } else if (!std::strcmp(a, "etypes")) {
      if (++iarg >= narg - 1) throw ConpError(CONP_ERR_ARG, "Invalid fix conp command (Insufficient input entries for etypes)");
      out->eletypenum = inumeric(arg[iarg], "fix conp etypes");
      if (out->eletypenum < 0 || out->eletypenum > 32) throw ConpError(CONP_ERR_ARG, "Invalid fix conp command (etypes count out of range)");
      for (int i = 0; i < out->eletypenum; ++i) {
        if (++iarg >= narg) throw ConpError(CONP_ERR_ARG, "Invalid fix conp command (Insufficient input entries for etypes)");
        out->eletypes[i] = inumeric(arg[iarg], "fix conp etypes");
      }
      for (int i = 0; i < out->eletypenum; ++i)
        if (out->eletypes[i] > ntypes) throw ConpError(CONP_ERR_ARG, "Invalid fix conp command (Invalid atom type in etypes)");
      out->smartlist = 1;
    } else if (!std::strcmp(a, "zneutr")) out->zneutr = 1;
    else if (!std::strcmp(a, "matout")) out->matout = 1;
    else if (!std::strcmp(a, "pppm")) out->pppm = 1;
    else if (!std::strcmp(a, "split")) out->split = 1;
    else if (!std::strcmp(a, "qinit")) out->qinit = 1;
    else if (!std::strcmp(a, "himem")) out->lowmem = 0;
    else if (!std::strcmp(a, "nonneutral")) out->nullneutral = 0;
    else if (!std::strcmp(a, "ehgo")) out->ehgo = 1;
    else if (!std::strcmp(a, "cg")) {          // new: v1.1 cannot reach its CG solver; v0.9 had a selector
      out->minimizer = CONP_SOLVER_CG;
      while (iarg + 2 < narg + 0 && (!std::strcmp(arg[iarg + 1], "maxiter") || !std::strcmp(arg[iarg + 1], "tol"))) {
        if (!std::strcmp(arg[iarg + 1], "maxiter")) out->maxiter = inumeric(arg[iarg + 2], "fix conp cg maxiter");
        else out->tolerance = numeric(arg[iarg + 2], "fix conp cg tol");
        iarg += 2;
      }
    } else {
      throw ConpError(CONP_ERR_ARG, std::string("Invalid fix conp commmand (unknown option: ") + a + ")");
    }
  }
  CONP_GUARD_END
}

int conp_fix_create(const conp_fix_args *args, const conp_env *env, conp_fix **out) {
  CONP_GUARD_BEGIN
  if (!args || !env || !out) throw ConpError(CONP_ERR_ARG, "null argument");
  if (args->split) throw ConpError(CONP_ERR_ARG, "split provider is not available in the HIP provider (experimental in the reference)");
  if (env->nranks < 1 || env->rank < 0 || env->rank >= env->nranks) throw ConpError(CONP_ERR_ARG, "bad rank/nranks");
  std::unique_ptr<conp_fix> f(new conp_fix());
  f->args = *args;
  f->env = *env;
  const size_t nc = (size_t)(env->ntypes + 1) * (env->ntypes + 1);
  f->cutsq_h.assign(env->cutsq, env->cutsq + nc);
  f->env.cutsq = nullptr;
  f->init_device();
  // `himem` (fix_conp.cpp:168 -> lowmemflag = false) makes KSpaceModuleEwald keep pre-scaled csk / snk[Ne][K] tables instead of
  // expanding them per atom (km_ewald.cpp:263-268, 498-506): a memory / time trade of the CPU provider.  This provider has ONE
  // formulation (phases regenerated on the matrix cores, nothing of size Ne x K stored), so the keyword selects nothing -- said
  // once where the reference prints its own notes (utils::logmesg), not silently.
  if (!args->lowmem)
    f->mesgf("conp/hip: keyword himem has no effect: the HIP provider keeps no Ne x K phase table in either mode\n");
  *out = f.release();
  CONP_GUARD_END
}

void conp_fix_destroy(conp_fix *fix) { delete fix; }

int conp_fix_init_list(conp_fix *f, int which, const conp_neighlist *l) {
  CONP_GUARD_BEGIN
  if (!f || !l) throw ConpError(CONP_ERR_ARG, "null argument");
  f->drop_graph();
  ListView v; v.inum = l->inum; v.ilist = l->ilist; v.numneigh = l->numneigh; v.first = l->first; v.neigh = l->neigh;
  if (which == 0 || which == 2) { f->alist = v; f->have_alist = true; }
  if (which == 1 || which == 2) { f->blist = v; f->have_blist = true; }
  if (which < 0 || which > 2) throw ConpError(CONP_ERR_ARG, "init_list: which must be 0, 1 or 2");
  CONP_GUARD_END
}

int conp_fix_setup_post_neighbor(conp_fix *f, const conp_atoms *at) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->linalg_init(at);
  f->post_neighbor(at);
  CONP_GUARD_END
}

int conp_fix_post_neighbor(conp_fix *f, const conp_atoms *at) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->post_neighbor(at);
  CONP_GUARD_END
}

int conp_fix_linalg_setup(conp_fix *f, const conp_atoms *at) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->linalg_setup(at);
  CONP_GUARD_END
}

int conp_fix_setup_pre_force(conp_fix *f, const conp_atoms *at, int64_t ntimestep, double potdiff) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->linalg_setup(at);
  f->pre_force(at, ntimestep, potdiff);
  CONP_GUARD_END
}

int conp_fix_pre_force(conp_fix *f, const conp_atoms *at, int64_t ntimestep, double potdiff) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->pre_force(at, ntimestep, potdiff);
  CONP_GUARD_END
}

double conp_fix_compute_scalar(const conp_fix *cf) {
  conp_fix *f = const_cast<conp_fix *>(cf);
  if (f->left_stale) {          // device-resident updates do not bring the scalar over every step: form and fetch it now
    try { (void)hipSetDevice(f->env.device); f->finish_scalar(f->left_potdiff); } catch (...) {}
  }
  return f->scalar_output;
}

int conp_fix_modify_param(conp_fix *f, int narg, const char *const *arg, int *consumed) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  const int n = f->modify_param(narg, arg);
  // the reference rebuilds its per-type tables in FixConp::init() of every run (fix_conp.cpp:296-299): a fix_modify that arrives
  // after the first setup must reach the device tables too
  if (n > 0 && f->idx.initialised) { f->ehgo_setup_tables(); f->sync(); }
  if (consumed) *consumed = n;
  CONP_GUARD_END
}

int conp_fix_post_force(conp_fix *f, const conp_atoms *at, double *fo, double *ek, double *ec, double *vir) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  if (f->runstage < 2) throw ConpError(CONP_ERR_STATE, "post_force before setup");
  f->post_force(at, fo, ek, ec, vir);
  CONP_GUARD_END
}

int conp_fix_post_force_step(conp_fix *f, const conp_atoms *at, int64_t ntimestep, double *fo, double *ek, double *ec, double *vir) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  if (f->runstage < 2) throw ConpError(CONP_ERR_STATE, "post_force before setup");
  f->post_force(at, fo, ek, ec, vir, ntimestep);
  CONP_GUARD_END
}

int conp_fix_a_cal(conp_fix *f, const conp_atoms *at) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->a_cal(at);
  CONP_GUARD_END
}

int conp_fix_b_cal(conp_fix *f, const conp_atoms *at) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->b_cal(at);
  f->sync();
  CONP_GUARD_END
}

int conp_fix_equation_solve(conp_fix *f) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->equation_solve();
  CONP_GUARD_END
}

int conp_fix_update_charge(conp_fix *f, const conp_atoms *at, double potdiff) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->update_charge(at, potdiff);
  CONP_GUARD_END
}

int conp_km_conp_setup(conp_fix *f, double qsqsum, int64_t natoms) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->km_conp_setup(qsqsum, natoms);
  CONP_GUARD_END
}

int conp_km_a_cal(conp_fix *f, const conp_atoms *at, double *aaa) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  const int ne = f->idx.elenum_all;
  f->km_a_read(at);
  f->km_a_cal_device();
  const double MY_PIS = 1.77245385090551602729;
  if (!f->setup_sharded() || f->env.rank == 0)
    launch_a_diag_slab(f->stream, ne, f->kt.ug_tot - (2.0 / MY_PIS) * f->kt.g_ewald, 0.0, nullptr, f->kt.slabflag == 1,
                       12.56637061435917295384 / f->kt.volume, f->d_ele_z.p, f->d_A.p);
  f->allreduce_matrix(f->d_A.p, (size_t)ne * ne);
  HIP_TRY(hipMemcpyAsync(aaa, f->d_A.p, (size_t)ne * ne * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  f->sync();
  CONP_GUARD_END
}

int conp_km_b_cal(conp_fix *f, const conp_atoms *at, double *bbb) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  if (!f->tables_current) f->km_a_read(at);
  if (at->nlocal + at->nghost != f->nall) throw ConpError(CONP_ERR_STATE, "atom count changed without post_neighbor");
  if (f->elyte_list_stale(at)) { f->sync(); f->build_elyte_list(at); }
  f->upload_xq(at);
  if (f->decomposed) f->gather_elyte(at);
  f->b_cal_device(f->d_x.p, f->d_q.p, false);
  if (f->nccl || f->rc.active()) f->allreduce_b();
  HIP_TRY(hipMemcpyAsync(bbb, f->d_b, f->idx.elenum_all * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  f->sync();
  CONP_GUARD_END
}

int conp_fix_info(const conp_fix *f, conp_info *o) {
  CONP_GUARD_BEGIN
  std::memset(o, 0, sizeof(*o));
  o->elenum = f->idx.elenum; o->elenum_all = f->idx.elenum_all; o->elytenum = f->idx.elytenum;
  o->maxtag_all = f->idx.maxtag_all; o->runstage = f->runstage;
  o->kcount = f->kt.kcount; o->kcount_flat = f->kt.kcount_flat; o->kcount_expand = f->kt.kcount_expand;
  o->kxmax = f->kt.kxmax; o->kymax = f->kt.kymax; o->kzmax = f->kt.kzmax; o->kmax = f->kt.kmax; o->kmax3d = f->kt.kmax3d;
  for (int i = 0; i < 7; ++i) o->kcount_dims[i] = f->kt.kcount_dims[i];
  o->cg_iterations = f->cg_iterations;
  o->n_zclasses = f->nzc;
  for (int i = 0; i < 3; ++i) o->unitk[i] = f->kt.unitk[i];
  o->volume = f->kt.volume; o->gsqmx = f->kt.gsqmx; o->ug_tot = f->kt.ug_tot; o->totsetq = f->totsetq;
  o->scalar_output = f->scalar_output; o->totinve = f->totinve; o->slabcorr = f->slabcorr;
  int64_t nbp = f->n_b_pairs;
  if (f->np_pending) { HIP_TRY(hipStreamSynchronize(f->stream)); nbp = *f->h_np; }     // (the count of a regrouping still in flight)
  o->n_blist_pairs = nbp;
  o->inverse_path = f->inverse_path; o->inverse_retries = f->inverse_retries; o->pppm_elyte_spreads = f->pp_elyte_spreads;
  const bool zn_on = f->zn_use() || f->zn_gen_use();
  o->zn_cols = zn_on ? 16 * f->zn_ncf : 0; o->zn_grid = zn_on ? f->zn_n : 0; o->zn_rows = zn_on ? 128 * (int)f->own_rt_h.size() : 0; o->n_alist_pairs = f->arows.npairs(); o->n_elyte_charged = f->nl;
  CONP_GUARD_END
}

int conp_fix_get_ktables(const conp_fix *f, int *kx, int *ky, int *kz, double *ug, int *kxy_list, int *kz_list) {
  CONP_GUARD_BEGIN
  if (!f->kspace_ready) throw ConpError(CONP_ERR_STATE, "k tables not built");
  const size_t K = f->kt.kcount, E = f->kt.kcount_expand;
  if (kx) std::memcpy(kx, f->kt.kxvecs.data(), K * sizeof(int));
  if (ky) std::memcpy(ky, f->kt.kyvecs.data(), K * sizeof(int));
  if (kz) std::memcpy(kz, f->kt.kzvecs.data(), K * sizeof(int));
  if (ug) std::memcpy(ug, f->kt.ug.data(), K * sizeof(double));
  if (kxy_list) std::memcpy(kxy_list, f->kt.kxy_list.data(), E * sizeof(int));
  if (kz_list) std::memcpy(kz_list, f->kt.kz_list.data(), E * sizeof(int));
  CONP_GUARD_END
}

int conp_fix_get_maps(const conp_fix *f, int *ele2tag, int *ele2eleall, int *eleall2tag, int *eleall2ele,
                      int *elecheck_eleall, int *elebuf2eleall, int *tag2eleall) {
  CONP_GUARD_BEGIN
  const EleIndex &x = f->idx;
  auto cp = [](int *dst, const std::vector<int> &v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(int)); };
  cp(ele2tag, x.ele2tag); cp(ele2eleall, x.ele2eleall); cp(eleall2tag, x.eleall2tag); cp(eleall2ele, x.eleall2ele);
  cp(elecheck_eleall, x.elecheck_eleall); cp(elebuf2eleall, x.elebuf2eleall); cp(tag2eleall, x.tag2eleall);
  CONP_GUARD_END
}

int conp_fix_get_matrix(conp_fix *f, double *aaa) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->unshard_rows();             // several ranks keep only their rows of S after the setup: collective re-assembly
  const size_t ne = f->idx.elenum_all;
  HIP_TRY(hipMemcpyAsync(aaa, f->d_A.p, ne * ne * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  f->sync();
  CONP_GUARD_END
}

int conp_fix_set_matrix(conp_fix *f, const double *aaa, int runstage) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  if (f->s_sharded) { f->d_Srows.release(); f->s_sharded = false; }
  const size_t ne = f->idx.elenum_all;
  f->d_A.reserve(ne * ne);
  ++f->s_generation;
  HIP_TRY(hipMemcpyAsync(f->d_A.p, aaa, ne * ne * sizeof(double), hipMemcpyHostToDevice, f->stream));
  f->sync();
  f->runstage = runstage;
  CONP_GUARD_END
}

int conp_fix_get_vectors(conp_fix *f, double *bbb_all, double *eleallq, double *elesetq) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  const size_t nb = f->idx.elenum_all * sizeof(double);
  if (bbb_all) HIP_TRY(hipMemcpyAsync(bbb_all, f->d_b, nb, hipMemcpyDeviceToHost, f->stream));
  if (eleallq) HIP_TRY(hipMemcpyAsync(eleallq, f->d_eleallq, nb, hipMemcpyDeviceToHost, f->stream));
  if (elesetq) HIP_TRY(hipMemcpyAsync(elesetq, f->d_elesetq.p, nb, hipMemcpyDeviceToHost, f->stream));
  f->sync();
  CONP_GUARD_END
}

int conp_fix_get_sfac(conp_fix *f, double *sr, double *si) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->refresh_structure_factors();
  const int K = f->kt.kcount;
  f->d_sfr.reserve(K); f->d_sfi.reserve(K);
  launch_sfac_gather(f->stream, K, f->plan.C_pad, KPlan::PT, f->d_sf_row_a.p, f->d_sf_col_c.p, f->d_k_sign.p, f->d_G.p,
                     f->d_sfr.p, f->d_sfi.p);
  HIP_TRY(hipMemcpyAsync(sr, f->d_sfr.p, K * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(hipMemcpyAsync(si, f->d_sfi.p, K * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  f->sync();
  CONP_GUARD_END
}

int conp_fix_get_ele_trig(conp_fix *f, double *csk, double *snk) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  if (!f->tables_current) throw ConpError(CONP_ERR_STATE, "electrode tables not built (a_cal / a_read first)");
  // a read-back of the DEVICE tables into the reference's layout csk / snk[i][kflat] (km_ewald.cpp:426-477): x axis, y axis, z axis,
  // then the (kx, +-ky) pairs -- what the kernels of every later step actually read
  const KTables &kt = f->kt;
  const KPlan &pl = f->plan;
  const int ne = f->idx.elenum_all, np_ = f->ne_pad, kflat = kt.kcount_flat;
  const int d0 = kt.kcount_dims[0], d1 = kt.kcount_dims[1], d2 = kt.kcount_dims[2];
  std::vector<double2> xe((size_t)(d0 + 2) * np_), ye((size_t)(d1 + 1) * np_);
  std::vector<double> rp((size_t)pl.R_pad * np_), tz((size_t)pl.C_pad * np_);
  HIP_TRY(hipMemcpyAsync(xe.data(), f->d_Xe.p, xe.size() * sizeof(double2), hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(hipMemcpyAsync(ye.data(), f->d_Ye.p, ye.size() * sizeof(double2), hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(hipMemcpyAsync(rp.data(), f->d_Rp.p, rp.size() * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  HIP_TRY(hipMemcpyAsync(tz.data(), f->d_Tz.p, tz.size() * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  f->sync();
  for (int i = 0; i < ne; ++i) {
    double *c = csk + (size_t)i * kflat, *s = snk + (size_t)i * kflat;
    for (int k = 1; k <= d0; ++k) { c[k - 1] = xe[(size_t)k * np_ + i].x; s[k - 1] = xe[(size_t)k * np_ + i].y; }
    for (int k = 1; k <= d1; ++k) { c[d0 + k - 1] = ye[(size_t)k * np_ + i].x; s[d0 + k - 1] = ye[(size_t)k * np_ + i].y; }
    for (int m = 1; m <= d2; ++m) {
      c[d0 + d1 + m - 1] = tz[(size_t)pl.col_c(m) * np_ + i];
      s[d0 + d1 + m - 1] = tz[(size_t)pl.col_s(m) * np_ + i];
    }
    for (int fl = d0 + d1 + d2; fl < kflat; ++fl) {
      const int p = pl.flat2p[fl];
      c[fl] = rp[(size_t)pl.row_a(p) * np_ + i];
      s[fl] = rp[(size_t)pl.row_b(p) * np_ + i];
    }
  }
  CONP_GUARD_END
}

int conp_inv_project(conp_fix *f, int n, double *aaa, int nullneutral, int zneutr, const double *eleallz, double zhalf,
                     double *totinve_out) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  DevBuf<double> dA;
  dA.upload(aaa, (size_t)n * n, f->stream);
  const int save_nn = f->args.nullneutral, save_zn = f->args.zneutr;
  f->args.nullneutral = nullneutral; f->args.zneutr = zneutr;
  std::vector<double> z(eleallz ? eleallz : aaa, (eleallz ? eleallz : aaa) + n);
  try { f->inv_project_device(n, dA.p, &z, zhalf); } catch (...) { f->args.nullneutral = save_nn; f->args.zneutr = save_zn; throw; }
  f->args.nullneutral = save_nn; f->args.zneutr = save_zn;
  HIP_TRY(hipMemcpyAsync(aaa, dA.p, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  f->sync();
  if (totinve_out) *totinve_out = f->totinve;
  CONP_GUARD_END
}

int conp_fix_write_matrix_file(conp_fix *f, const char *path, int which) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  if (f->runstage < 1) throw ConpError(CONP_ERR_STATE, "no matrix yet");
  f->write_matrix_file(path, which);
  CONP_GUARD_END
}

int conp_fix_read_matrix_file(conp_fix *f, const conp_atoms *at, const char *path) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  if (!f->idx.initialised || f->idx.elenum_all == 0) throw ConpError(CONP_ERR_STATE, "read_matrix_file before setup_post_neighbor");
  if (f->args.a_matrix_f == 0) f->args.a_matrix_f = 1;
  f->a_read_file(at, path);
  CONP_GUARD_END
}

int conp_invert(conp_fix *f, int n, double *aaa) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  DevBuf<double> dA;
  dA.upload(aaa, (size_t)n * n, f->stream);
  f->invert_device(n, dA.p);
  HIP_TRY(hipMemcpyAsync(aaa, dA.p, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  f->sync();
  CONP_GUARD_END
}

int conp_host_ktables(double g_ewald, double accuracy, double slab_volfactor, int slabflag, double xprd, double yprd, double zprd,
                      double qsqsum, int64_t natoms, double qqrd2e, double dielectric, int *info, int *kx, int *ky, int *kz,
                      double *ug, int *kxy_list, int *kz_list, int *plan_p, int *plan_m, int *plan_sign) {
  CONP_GUARD_BEGIN
  KTables kt;
  kt.build(g_ewald, accuracy, slab_volfactor, slabflag, xprd, yprd, zprd, qsqsum, natoms, qqrd2e, dielectric);
  KPlan pl;
  pl.build(kt);
  if (info) {
    const int v[9] = {kt.kcount, kt.kcount_flat, kt.kcount_expand, kt.kxmax, kt.kymax, kt.kzmax, kt.kmax, kt.kmax3d, 0};
    for (int i = 0; i < 8; ++i) info[i] = v[i];
    for (int i = 0; i < 7; ++i) info[8 + i] = kt.kcount_dims[i];
    info[15] = pl.np;
  }
  const size_t K = kt.kcount, E = kt.kcount_expand;
  if (kx) std::memcpy(kx, kt.kxvecs.data(), K * sizeof(int));
  if (ky) std::memcpy(ky, kt.kyvecs.data(), K * sizeof(int));
  if (kz) std::memcpy(kz, kt.kzvecs.data(), K * sizeof(int));
  if (ug) std::memcpy(ug, kt.ug.data(), K * sizeof(double));
  if (kxy_list) std::memcpy(kxy_list, kt.kxy_list.data(), E * sizeof(int));
  if (kz_list) std::memcpy(kz_list, kt.kz_list.data(), E * sizeof(int));
  if (plan_p) std::memcpy(plan_p, pl.k_p.data(), K * sizeof(int));
  if (plan_m) std::memcpy(plan_m, pl.k_m.data(), K * sizeof(int));
  if (plan_sign) std::memcpy(plan_sign, pl.k_sign.data(), K * sizeof(int));
  CONP_GUARD_END
}

int conp_host_index(int n0, const int *tag0, const int *echeck0, int n1, const int *tag1, const int *echeck1, int *sizes,
                    int *ele2tag, int *ele2eleall, int *eleall2tag, int *eleall2ele, int *elebuf2eleall, int *tag2eleall) {
  CONP_GUARD_BEGIN
  EleIndex x;
  x.linalg_init(n0, tag0);
  x.post_neighbor(n0, tag0, echeck0, nullptr);
  if (n1 > 0) x.post_neighbor(n1, tag1, echeck1, nullptr);
  if (sizes) { sizes[0] = x.elenum; sizes[1] = x.elenum_all; sizes[2] = x.elytenum; sizes[3] = x.maxtag_all; }
  auto cp = [](int *dst, const std::vector<int> &v) { if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(int)); };
  cp(ele2tag, x.ele2tag); cp(ele2eleall, x.ele2eleall); cp(eleall2tag, x.eleall2tag); cp(eleall2ele, x.eleall2ele);
  cp(elebuf2eleall, x.elebuf2eleall); cp(tag2eleall, x.tag2eleall);
  CONP_GUARD_END
}

int64_t conp_host_pair_rows(int which, const conp_neighlist *l, const conp_atoms *at, int newton, int *row_ptr, int *ele_atom,
                            int *oth_atom, int *col) {
  try {
    EleIndex x;
    x.linalg_init(at->nlocal, at->tag);
    x.post_neighbor(at->nlocal, at->tag, at->echeck, nullptr);
    ListView v; v.inum = l->inum; v.ilist = l->ilist; v.numneigh = l->numneigh; v.first = l->first; v.neigh = l->neigh;
    auto cp = [](int *dst, const std::vector<int> &s) { if (dst && !s.empty()) std::memcpy(dst, s.data(), s.size() * sizeof(int)); };
    if (which == 2) {
      std::vector<int> pi, pj;
      build_pf_pairs(v, at->echeck, pi, pj);
      cp(ele_atom, pi); cp(oth_atom, pj);
      return (int64_t)pi.size();
    }
    PairRows r;
    if (which == 1) build_b_rows(v, at->nlocal, at->tag, at->echeck, x, newton != 0, r);
    else build_a_rows(v, at->nlocal, at->tag, at->echeck, x, newton != 0, r);
    cp(row_ptr, r.row_ptr); cp(ele_atom, r.ele_atom); cp(oth_atom, r.oth_atom); cp(col, r.col);
    return r.npairs();
  } catch (const std::exception &e) { g_last_error = e.what(); return -1; }
}

// ---- PPPM coupling beyond b, compute potential/atom --------------------------------------------------------------------------
namespace {
void need_pppm(conp_fix *f) {
  if (!f->args.pppm || f->dpppm.nfft <= 0)
    throw ConpError(CONP_ERR_STATE, "Compute requires a compatible KSpace provider like pppm/conp");   // compute_potential_atom.cpp:110
}
// Spatially decomposed ranks: the mesh potentials and density bricks are those of ALL atoms.  Every rank gathers the charged owned
// atoms of all ranks -- (x, q, kind), ONE conp_comm gather per query -- and spreads them onto its own copy of the whole mesh: a
// REPLICATED mesh (the reference's ranks own bricks of it and exchange ghost planes, pppm_conp.cpp:114,122 GridComm; the mesh of
// the `pppm` mode is 10^5..10^6 points), so the collective call gives every rank the same brick and the potentials of its own
// atoms.  The gathered atoms live in buffers of their OWN (d_pp_xg / d_pp_qg / d_pp_iota): the per-update electrolyte gather of the
// decomposed b path keeps d_xg / d_qg / d_iota to itself.  counts[kind] / first[kind]: the atoms of a kind are one run of d_pp_iota
// (0 electrolyte, 1 electrode); kind 2 = all = the whole list.
struct PpGather { int n[3], first[3]; };
PpGather pppm_gather_all(conp_fix *f, const conp_atoms *at) {
  std::vector<double> pack;
  for (int i = 0; i < at->nlocal; ++i) {
    if (at->q[i] == 0) continue;
    pack.push_back(at->x[3 * (size_t)i]); pack.push_back(at->x[3 * (size_t)i + 1]); pack.push_back(at->x[3 * (size_t)i + 2]);
    pack.push_back(at->q[i]); pack.push_back(at->echeck[i] != 0 ? 1.0 : 0.0);
  }
  std::vector<int> counts(f->env.nranks, 0);
  f->rc.allgather_int((int)(pack.size() / 5), counts.data());
  int n = 0;
  for (int v : counts) n += v;
  std::vector<double> all((size_t)std::max(n, 1) * 5), xs((size_t)std::max(n, 1) * 3), qs(std::max(n, 1));
  if (pack.empty()) pack.resize(5);
  f->rc.gatherv(pack.data(), counts, 5, all.data());
  std::vector<int> iota;
  iota.reserve(std::max(n, 1));
  PpGather g{};
  for (int kind = 0; kind < 2; ++kind) {
    g.first[kind] = (int)iota.size();
    for (int k = 0; k < n; ++k) if ((all[5 * (size_t)k + 4] != 0.0) == (kind == 1)) iota.push_back(k);
    g.n[kind] = (int)iota.size() - g.first[kind];
  }
  g.first[2] = 0; g.n[2] = n;
  for (int k = 0; k < n; ++k) {
    xs[3 * (size_t)k] = all[5 * (size_t)k]; xs[3 * (size_t)k + 1] = all[5 * (size_t)k + 1]; xs[3 * (size_t)k + 2] = all[5 * (size_t)k + 2];
    qs[k] = all[5 * (size_t)k + 3];
  }
  if (iota.empty()) iota.push_back(0);
  f->d_pp_iota.upload(iota, f->stream);
  f->d_pp_xg.upload(xs, f->stream); f->d_pp_qg.upload(qs, f->stream);
  f->sync();                       // (the host vectors go out of scope)
  return g;
}
// index lists of the owned atoms that carry charge, by kind (0 electrolyte, 1 electrode, 2 all), and x, q on the device
int pppm_list(conp_fix *f, const conp_atoms *at, int kind, DevBuf<int> &d_idx) {
  std::vector<int> idx;
  for (int i = 0; i < at->nlocal; ++i) {
    if (at->q[i] == 0) continue;
    if (kind == 0 && at->echeck[i] != 0) continue;
    if (kind == 1 && at->echeck[i] == 0) continue;
    idx.push_back(i);
  }
  const int n = (int)idx.size();
  if (idx.empty()) idx.push_back(0);
  d_idx.upload(idx, f->stream);
  return n;
}
void pppm_upload(conp_fix *f, const conp_atoms *at) {
  if (at->nlocal + at->nghost != f->nall) throw ConpError(CONP_ERR_STATE, "atom count changed without post_neighbor");
  f->resident_step = -1;
  f->upload_xq_n(at, f->nall);
}
// u_brick of the total density into d_pp_re (what PPPM::compute leaves there with per-atom energies on).  COLLECTIVE under ranks.
void pppm_total_potential(conp_fix *f, const conp_atoms *at) {
  DevBuf<int> d_idx;
  pppm_upload(f, at);
  f->d_pp_scratch.reserve(2048);
  if (f->decomposed) {
    const PpGather g = pppm_gather_all(f, at);
    launch_pppm_density(f->stream, f->dpppm, g.n[2], f->d_pp_iota.p, f->d_pp_xg.p, f->d_pp_qg.p, f->d_pp_re.p, f->d_pp_scratch.p);
  } else {
    const int n = pppm_list(f, at, 2, d_idx);
    launch_pppm_density(f->stream, f->dpppm, n, d_idx.p, f->d_x.p, f->d_q.p, f->d_pp_re.p, f->d_pp_scratch.p);
  }
  ++f->pp_elyte_spreads;
  launch_pppm_poisson(f->stream, f->dpppm, f->d_pp_re.p, f->d_pp_im.p); f->pp_im_clean = false;
  HIP_TRY(hipGetLastError());
  f->sync();                       // d_idx goes out of scope
  f->pp_u_valid = true;
}
}  // namespace

// PPPMCONP keeps the electrolyte brick of every b_cal for its make_rho override (pppm_conp.cpp:172-228, 434-450; elyte_mapped is
// reset by conp_pre_force, pppm_conp.h:42).  on != 0: from the next b_cal on the brick of the update is kept on the device
// (d_pp_elyte) and conp_pppm_make_rho adds the electrode brick to it instead of spreading the electrolyte again; the call itself
// drops whatever brick is cached (a new step begins).
int conp_pppm_keep_density(conp_fix *f, int on) {
  CONP_GUARD_BEGIN
  need_pppm(f);
  f->pp_keep = on != 0;
  f->pp_elyte_valid = false;
  f->pp_u_valid = false;
  CONP_GUARD_END
}

// The mesh potential of the total density -- what PPPM::compute leaves in u_brick when per-atom energies are tallied.  COLLECTIVE
// under ranks (one gather of all ranks' charged atoms, a replicated mesh solve); afterwards conp_pppm_compute_particle_potential is
// a rank-local stencil gather from the cached brick, like the reference's (pppm_conp.cpp:452-485), until the next update.
int conp_pppm_compute(conp_fix *f, const conp_atoms *at) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  need_pppm(f);
  pppm_total_potential(f, at);
  CONP_GUARD_END
}

int conp_pppm_make_rho(conp_fix *f, const conp_atoms *at, double *density, double *ele_density, double *elyte_density) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  need_pppm(f);
  pppm_upload(f, at);
  const size_t nf = (size_t)f->dpppm.nfft;
  f->d_pp_scratch.reserve(2048);
  f->d_pp_ele.reserve(nf);
  std::vector<double> e(nf), l(nf);
  DevBuf<int> d_idx0, d_idx1;
  PpGather g{};
  if (f->decomposed) g = pppm_gather_all(f, at);          // ONE gather, the atoms tagged with their kind
  {
    if (f->decomposed)
      launch_pppm_density(f->stream, f->dpppm, g.n[1], f->d_pp_iota.p + g.first[1], f->d_pp_xg.p, f->d_pp_qg.p, f->d_pp_ele.p, f->d_pp_scratch.p);
    else {
      const int n = pppm_list(f, at, 1, d_idx1);          // ele_make_rho (pppm_conp.cpp:385-426)
      launch_pppm_density(f->stream, f->dpppm, n, d_idx1.p, f->d_x.p, f->d_q.p, f->d_pp_ele.p, f->d_pp_scratch.p);
    }
    HIP_TRY(hipMemcpyAsync(e.data(), f->d_pp_ele.p, nf * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  }
  if (f->pp_elyte_valid && !f->decomposed) {
    // the brick b_cal left for this step (elyte_mapped, pppm_conp.cpp:437-442): no second spread of the electrolyte
    HIP_TRY(hipMemcpyAsync(l.data(), f->d_pp_elyte.p, nf * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  } else {
    if (f->decomposed)
      launch_pppm_density(f->stream, f->dpppm, g.n[0], f->d_pp_iota.p + g.first[0], f->d_pp_xg.p, f->d_pp_qg.p, f->d_pp_re.p, f->d_pp_scratch.p);
    else {
      const int n = pppm_list(f, at, 0, d_idx0);          // elyte_make_rho (:172-228)
      launch_pppm_density(f->stream, f->dpppm, n, d_idx0.p, f->d_x.p, f->d_q.p, f->d_pp_re.p, f->d_pp_scratch.p);
    }
    ++f->pp_elyte_spreads;
    f->pp_u_valid = false;                                 // (d_pp_re holds a density now)
    HIP_TRY(hipMemcpyAsync(l.data(), f->d_pp_re.p, nf * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  }
  f->sync();
  if (ele_density) std::memcpy(ele_density, e.data(), nf * sizeof(double));
  if (elyte_density) std::memcpy(elyte_density, l.data(), nf * sizeof(double));
  if (density) for (size_t k = 0; k < nf; ++k) density[k] = l[k] + e[k];        // make_rho override :434-450
  CONP_GUARD_END
}

int conp_pppm_compute_group_potential(conp_fix *f, const conp_atoms *at, const int *sel, double *recv) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  need_pppm(f);
  if (!sel || !recv) throw ConpError(CONP_ERR_ARG, "null argument");
  pppm_total_potential(f, at);
  std::vector<int> idx;
  for (int i = 0; i < at->nlocal; ++i) if (sel[i]) idx.push_back(i);
  if (idx.empty()) return CONP_OK;
  DevBuf<int> d_idx;
  DevBuf<double> d_out;
  d_idx.upload(idx, f->stream);
  d_out.reserve(at->nlocal);
  launch_pppm_probe(f->stream, f->dpppm, (int)idx.size(), d_idx.p, f->d_x.p, f->d_q.p, f->d_pp_re.p, 0.0, d_out.p);
  std::vector<double> out(at->nlocal);
  HIP_TRY(hipMemcpyAsync(out.data(), d_out.p, (size_t)at->nlocal * sizeof(double), hipMemcpyDeviceToHost, f->stream));
  f->sync();
  for (int i : idx) recv[i] = out[i];
  CONP_GUARD_END
}

// RANK-LOCAL like the reference's (pppm_conp.cpp:452-485; compute_potential_atom.cpp:168-174 calls it once per owned atom of the
// group, a different number of times on every rank): a stencil gather from the cached mesh potential.  Under several ranks the brick
// must have been formed by a collective entry since the last update (conp_pppm_compute, conp_pppm_compute_group_potential,
// conp_compute_potential_atom); one rank forms it on demand.
int conp_pppm_compute_particle_potential(conp_fix *f, const conp_atoms *at, int i, double *u) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  need_pppm(f);
  if (!u || i < 0 || i >= at->nlocal) throw ConpError(CONP_ERR_ARG, "atom index out of range");
  if (!f->pp_u_valid) {
    if (f->decomposed || f->env.nranks > 1)
      throw ConpError(CONP_ERR_STATE, "pppm/conp/hip: compute_particle_potential is rank-local -- under several MPI ranks the mesh potential "
                                      "has to be formed by a collective call first (conp_pppm_compute / compute_group_potential)");
    pppm_total_potential(f, at);
  }
  if (at->nlocal + at->nghost != f->nall) throw ConpError(CONP_ERR_STATE, "atom count changed without post_neighbor");
  // the atom's CURRENT position and charge (4 doubles), the brick as it is
  HIP_TRY(hipMemcpyAsync(f->d_x.p + 3 * (size_t)i, at->x + 3 * (size_t)i, 3 * sizeof(double), hipMemcpyHostToDevice, f->stream));
  HIP_TRY(hipMemcpyAsync(f->d_q.p + i, at->q + i, sizeof(double), hipMemcpyHostToDevice, f->stream));
  std::vector<int> idx(1, i);
  DevBuf<int> d_idx;
  DevBuf<double> d_out;
  d_idx.upload(idx, f->stream);
  d_out.reserve(at->nlocal);
  launch_pppm_probe(f->stream, f->dpppm, 1, d_idx.p, f->d_x.p, f->d_q.p, f->d_pp_re.p,
                    2.0 * f->env.g_ewald / 1.77245385090551602729, d_out.p);
  HIP_TRY(hipMemcpyAsync(u, d_out.p + i, sizeof(double), hipMemcpyDeviceToHost, f->stream));
  f->sync();
  CONP_GUARD_END
}

int conp_compute_potential_atom(conp_fix *f, const conp_atoms *at, const conp_neighlist *pl, const int *sel, const int *etasel,
                                const conp_potential_args *pa, double *potential) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  if (!at || !sel || !pa || !potential) throw ConpError(CONP_ERR_ARG, "null argument");
  if (pa->eta != 0.0 && !etasel) throw ConpError(CONP_ERR_ARG, "eta needs the eta_check selection");
  if (pa->kspaceflag) need_pppm(f);
  const int nall = at->nlocal + at->nghost;
  const int ntotal = at->nlocal + (f->env.newton_pair ? at->nghost : 0);
  pppm_upload(f, at);
  DevBuf<double> d_pot;
  DevBuf<int> d_sel, d_eta, d_il, d_nn, d_first, d_neigh, d_idx;
  d_pot.reserve(nall); d_pot.zero(f->stream);
  d_sel.upload(sel, nall, f->stream);
  std::vector<int> ez(nall, 0);
  d_eta.upload(etasel ? etasel : ez.data(), nall, f->stream);
  if (pa->pairflag) {
    if (!pl) throw ConpError(CONP_ERR_ARG, "the pair part needs the pair style's neighbor list");
    size_t nneigh = 0;
    for (int ii = 0; ii < pl->inum; ++ii) nneigh = std::max(nneigh, (size_t)pl->first[pl->ilist[ii]] + (size_t)pl->numneigh[pl->ilist[ii]]);
    d_il.upload(pl->ilist, (size_t)pl->inum, f->stream); d_nn.upload(pl->numneigh, nall, f->stream);
    d_first.upload(pl->first, nall, f->stream); d_neigh.upload(pl->neigh, std::max<size_t>(nneigh, 1), f->stream);
    launch_potential_pair(f->stream, pl->inum, d_il.p, d_nn.p, d_first.p, d_neigh.p, at->nlocal, f->env.newton_pair, f->d_x.p,
                          f->d_q.p, f->d_type.p, d_sel.p, d_eta.p, f->env.ntypes, f->d_cutsq.p, f->real_params().cut_coulsq,
                          f->env.g_ewald, pa->eta, d_pot.p);
  }
  std::vector<double> pot(nall, 0.0), uk(at->nlocal, 0.0);
  if (pa->kspaceflag) {
    std::vector<int> all;
    for (int i = 0; i < at->nlocal; ++i) if (at->q[i] != 0) all.push_back(i);
    const int n = (int)all.size();
    if (all.empty()) all.push_back(0);
    d_idx.upload(all, f->stream);
    f->d_pp_scratch.reserve(2048);
    if (f->decomposed) {           // all ranks' charged atoms on this rank's copy of the mesh (pppm_gather_all)
      const PpGather g = pppm_gather_all(f, at);
      launch_pppm_density(f->stream, f->dpppm, g.n[2], f->d_pp_iota.p, f->d_pp_xg.p, f->d_pp_qg.p, f->d_pp_re.p, f->d_pp_scratch.p);
    } else
      launch_pppm_density(f->stream, f->dpppm, n, d_idx.p, f->d_x.p, f->d_q.p, f->d_pp_re.p, f->d_pp_scratch.p);
    launch_pppm_poisson(f->stream, f->dpppm, f->d_pp_re.p, f->d_pp_im.p); f->pp_im_clean = false;
    ++f->pp_elyte_spreads; f->pp_u_valid = true;
    std::vector<int> idx;
    for (int i = 0; i < at->nlocal; ++i) if (sel[i]) idx.push_back(i);
    DevBuf<int> d_pidx;
    DevBuf<double> d_uk;
    d_uk.reserve(std::max(at->nlocal, 1));
    if (!idx.empty()) {
      d_pidx.upload(idx, f->stream);
      launch_pppm_probe(f->stream, f->dpppm, (int)idx.size(), d_pidx.p, f->d_x.p, f->d_q.p, f->d_pp_re.p,
                        2.0 * f->env.g_ewald / 1.77245385090551602729, d_uk.p);
      HIP_TRY(hipMemcpyAsync(uk.data(), d_uk.p, (size_t)at->nlocal * sizeof(double), hipMemcpyDeviceToHost, f->stream));
    }
    HIP_TRY(hipMemcpyAsync(pot.data(), d_pot.p, (size_t)nall * sizeof(double), hipMemcpyDeviceToHost, f->stream));
    f->sync();
    const double MY_PIS = 1.77245385090551602729;
    for (int i : idx) {                                                           // :165-175
      pot[i] -= uk[i];
      if (pa->eta != 0.0 && etasel[i]) pot[i] += pa->eta * at->q[i] * std::sqrt(2.0) / MY_PIS;
    }
    if (f->env.slabflag) {                                                        // slabcorr :323-345
      const double volume = f->env.xprd * f->env.yprd * f->env.zprd * f->env.slab_volfactor;
      const double pi2vol = 2 * 3.14159265358979323846 / volume;
      double slabcorr = 0.0, qsum = 0.0;
      for (int i = 0; i < at->nlocal; ++i) { slabcorr += 2 * pi2vol * at->q[i] * at->x[3 * (size_t)i + 2]; qsum += at->q[i]; }
      { double two[2] = {slabcorr, qsum}; f->rc.sum(two, 2); slabcorr = two[0]; qsum = two[1]; }      // MPI_Allreduce :331, :337
      for (int i = 0; i < at->nlocal; ++i)
        if (sel[i]) {
          const double z = at->x[3 * (size_t)i + 2];
          pot[i] += z * slabcorr;
          if (pa->qsumflag) pot[i] -= pi2vol * qsum * z * z;
        }
    }
  } else {
    HIP_TRY(hipMemcpyAsync(pot.data(), d_pot.p, (size_t)nall * sizeof(double), hipMemcpyDeviceToHost, f->stream));
    f->sync();
  }
  const double evs = f->env.qqr2e / f->env.qe2f;                                  // :99, :214
  for (int i = 0; i < ntotal; ++i) potential[i] = pot[i] * evs;
  CONP_GUARD_END
}

int conp_fix_set_comm(conp_fix *f, const conp_comm *comm) {
  CONP_GUARD_BEGIN
  if (!f || !comm) throw ConpError(CONP_ERR_ARG, "null argument");
  if (f->idx.initialised) throw ConpError(CONP_ERR_STATE, "conp_fix_set_comm after setup_post_neighbor");
  if (comm->nranks < 1 || comm->rank < 0 || comm->rank >= comm->nranks) throw ConpError(CONP_ERR_ARG, "bad rank/nranks");
  if (comm->nranks > 1 && (!comm->allreduce_sum || !comm->allreduce_max_int || !comm->allgather_int || !comm->allgatherv))
    throw ConpError(CONP_ERR_ARG, "conp_comm: every callback is needed with more than one rank");
  f->rc.c = *comm; f->rc.have = true; f->rc.rank_ = comm->rank; f->rc.nranks_ = comm->nranks;
  f->env.rank = comm->rank; f->env.nranks = comm->nranks;
  f->decomposed = true;
  CONP_GUARD_END
}

int conp_rccl_unique_id(void *id_out) {
  CONP_GUARD_BEGIN
  if (!id_out) throw ConpError(CONP_ERR_ARG, "null argument");
  static_assert(sizeof(ncclUniqueId) == CONP_RCCL_ID_BYTES, "ncclUniqueId size");
  g_rccl.load();
  ncclUniqueId id;
  g_rccl.ok(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(id_out, &id, sizeof(id));
  CONP_GUARD_END
}

int conp_fix_comm_init_rccl(conp_fix *f, const void *id_in) {
  CONP_GUARD_BEGIN
  if (!f || !id_in) throw ConpError(CONP_ERR_ARG, "null argument");
  if (f->nccl) throw ConpError(CONP_ERR_STATE, "RCCL communicator already initialised");
  f->drop_graph();
  g_rccl.load();
  ncclUniqueId id;
  std::memcpy(&id, id_in, sizeof(id));
  g_rccl.ok(g_rccl.CommInitRank(&f->nccl, f->env.nranks, id, f->env.rank), "ncclCommInitRank");
  // results of the in-library collectives land in the library's own vectors
  f->d_b = f->d_b_own.p; f->d_eleallq = f->d_eleallq_own.p;
  f->b_bound = f->q_bound = false;
  CONP_GUARD_END
}

// 0 when librccl can be loaded and has every entry point the library calls.  Cheap and local (no communication): hosts call it on
// every rank and AGREE on the answer (MPI_Allreduce MIN, torch.distributed ...) BEFORE the collective conp_fix_comm_init_rccl,
// so that a rank without RCCL cannot leave its partners waiting inside ncclCommInitRank.
int conp_rccl_available(void) {
  try { g_rccl.load(); } catch (...) { return CONP_ERR_NO_DEVICE; }
  return CONP_OK;
}

// Gives the communicator back (all ranks, together: after an initialisation that failed on some rank, the host falls back to
// doing the exchanges itself).  The library's vectors stay its own.
int conp_fix_comm_destroy_rccl(conp_fix *f) {
  CONP_GUARD_BEGIN
  if (!f) throw ConpError(CONP_ERR_ARG, "null argument");
  f->drop_graph();
  if (f->nccl) { f->sync(); (void)g_rccl.CommDestroy(f->nccl); f->nccl = nullptr; }
  CONP_GUARD_END
}

int conp_fix_set_stream(conp_fix *f, void *s) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->sync();
  if (f->own_stream && f->stream) { (void)hipStreamDestroy(f->stream); f->own_stream = false; }
  f->stream = static_cast<hipStream_t>(s);
  CONP_GUARD_END
}

int conp_fix_bind_device_buffers(conp_fix *f, double *d_b, double *d_q) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->sync();
  if (f->nccl && d_q) throw ConpError(CONP_ERR_STATE, "with an RCCL communicator the library gathers q into its own buffer "
                                                      "(nranks * ceil(Ne / nranks) entries); read it with conp_fix_get_vectors");
  const size_t nb = f->idx.elenum_all * sizeof(double);
  if (d_b) { if (f->d_b && nb) HIP_TRY(hipMemcpy(d_b, f->d_b, nb, hipMemcpyDeviceToDevice)); f->d_b = d_b; f->b_bound = true; }
  if (d_q) { if (f->d_eleallq && nb) HIP_TRY(hipMemcpy(d_q, f->d_eleallq, nb, hipMemcpyDeviceToDevice)); f->d_eleallq = d_q; f->q_bound = true; }
  CONP_GUARD_END
}

int conp_fix_row_range(const conp_fix *f, int *r0, int *r1) {
  CONP_GUARD_BEGIN
  *r0 = f->row0; *r1 = f->row1;
  CONP_GUARD_END
}

int conp_fix_b_cal_device(conp_fix *f, const double *dx, const double *dq) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  if (f->runstage < 2) throw ConpError(CONP_ERR_STATE, "b_cal_device before setup");
  f->b_cal_device(dx, dq, true);
  CONP_GUARD_END
}

int conp_fix_solve_device(conp_fix *f, double potdiff) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  (void)potdiff;
  f->solve_device();
  CONP_GUARD_END
}

int conp_fix_scatter_device(conp_fix *f, double *d_q_atoms, double potdiff) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->scatter_device(d_q_atoms, potdiff);
  CONP_GUARD_END
}

int conp_fix_pre_force_device(conp_fix *f, const double *dx, double *dq, double potdiff) {
  CONP_GUARD_BEGIN
  if (f->runstage < 2) throw ConpError(CONP_ERR_STATE, "pre_force_device before setup");
  (void)hipSetDevice(f->env.device);
  f->update_device(dx, dq, potdiff);
  CONP_GUARD_END
}

int conp_fix_write_timing(conp_fix *f) {
  CONP_GUARD_BEGIN
  f->collect_b_times();
  f->logf("B vector calculation time = %g\n", f->Btime);          // fix_conp.cpp:564-566
  f->logf("Coulomb calculation time = %g\n", f->Ctime);
  f->logf("Kspace calculation time = %g\n", f->Ktime);
  CONP_GUARD_END
}

const char *conp_fix_mesg_drain(conp_fix *f) {
  f->mesgdrain.swap(f->mesgbuf);
  f->mesgbuf.clear();
  return f->mesgdrain.c_str();
}

const char *conp_fix_log_drain(conp_fix *f) {
  f->logdrain.swap(f->logbuf);
  f->logbuf.clear();
  return f->logdrain.c_str();
}

int conp_fix_pin_host_arrays(conp_fix *f, const double *x, const double *q, int n) {
  CONP_GUARD_BEGIN
  if (!f) throw ConpError(CONP_ERR_ARG, "null argument");
  f->drop_graph();
  f->pin_host(x, q, n);
  CONP_GUARD_END
}

int conp_fix_unpin_host_arrays(conp_fix *f) {
  CONP_GUARD_BEGIN
  if (!f) throw ConpError(CONP_ERR_ARG, "null argument");
  f->drop_graph();
  f->unpin_host();
  CONP_GUARD_END
}

void *conp_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}

void conp_host_free(void *p) { if (p) (void)hipHostFree(p); }

// CONP_GUARD=1: reads back the guard zones around every device buffer of the library (all handles of this process).  Returns the
// number of damaged zones (0: every kernel so far stayed inside its buffers), -1 when guard zones are off; conp_last_error()
// then names the buffers' sizes.
int conp_debug_check_guards(void) {
  if (!GuardZones::on()) return -1;
  std::string what;
  const int bad = GuardZones::get().check(what);
  if (bad) g_last_error = "guard zones damaged:" + what;
  return bad;
}

int conp_fix_profile(conp_fix *f, int enable) {
  CONP_GUARD_BEGIN
  f->drop_graph();
  f->sync();
  f->prof.reset();
  f->prof.on = enable != 0;
  f->prof.dominant_only = enable == 2;
  CONP_GUARD_END
}

int conp_fix_profile_read(conp_fix *f, int *nk, const char **names, double *avg_ms, int *counts) {
  CONP_GUARD_BEGIN
  f->sync();
  f->prof.collect();
  f->prof.names_keep = f->prof.order;
  int n = 0;
  for (const auto &nm : f->prof.names_keep) {
    if (n >= 16) break;
    const auto &e = f->prof.acc[nm];
    names[n] = f->prof.names_keep[n].c_str();
    avg_ms[n] = e.second ? e.first / e.second : 0.0;
    counts[n] = e.second;
    ++n;
  }
  *nk = n;
  CONP_GUARD_END
}

}  // extern "C"
