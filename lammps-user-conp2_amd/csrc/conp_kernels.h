// Launch wrappers of the gfx950 kernels (conp_kernels.hip).  All pointers are device pointers unless noted.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/conp_hip.h"      // CONP_PATH_* (the test hooks of the C ABI)

namespace conp {

// Alternative code paths.  (i) What the parity tests compare the default paths with is selected through the C ABI
// (conp_debug_set_paths, include/conp_hip.h: a process-wide bit mask, CONP_PATH_*) -- path_on().  (ii) Switches of decided
// experiments ("measured, not kept", DESIGN-LOG.md) and those that change the PHYSICS of a run (sk_gemm ablation, rank emulation,
// whole-tile sharding) exist in the diagnostic library only (-DCONP_DIAG, `make diag`): diag_switch() reads the environment
// there and folds to a null constant -- name and all -- in the product.  (iii) The product reads from the environment only
// operational knobs: CONP_GUARD, CONP_GRAPH, CONP_PANEL_SINGLE / _MAXG / _SPIN, CONP_HOST_THREADS, CONP_TIME_HOST / _REN.
bool path_on(unsigned bit);
int debug_sk_workgroups();
const char *env_knob(const char *name);      // operational knob: getenv + one line on stderr the first time a set knob is read
#ifdef CONP_DIAG
#define diag_switch(name) ::conp::env_knob(name)
#else
#define diag_switch(name) (static_cast<const char *>(nullptr))
#endif

struct DevPlan {              // device copy of KPlan geometry
  int np, nz, n_row_tiles, n_col_tiles, R_pad, C_pad, kxmax, kymax;
  const int *p_ikx, *p_iky, *p_sgn;   // [n_row_tiles*64] padded (padding: 0,0,0 -> sgn 0 marks "no vector")
  const int *nb_act;                  // [n_col_tiles][n_row_tiles] active 16-kz blocks of the column tile per row tile (KPlan::nba_rc)
  const double *wfull;                // [R_pad][C_pad]
};

// the z-window form (conp_zn.hip): item = (row tile: 64 planar vectors, chunk range [c0, c1) of the z-ordered electrolyte list, window
// origin g0 on the grid, slot of its piece [class][128 rows] in the pieces buffer)
struct ZnItem { int rt, c0, c1, g0, slot, paired; };     // paired: the row tile holds 32 whole (+ky, -ky) pairs (KPlan::paired_lo / _hi)
// what elyte_phase_kernel's z-axis threads need to write the window matrix of an update instead of the z phase seeds
struct ZnWindow { double *Bt; const int *g0c; int *flag; int ncol, n, W; double beta, gscale; };

// One sk_gemm segment: a BAND of planar vectors x one column tile x a chunk range [c0, c1) of 16 atoms.  A band = rf consecutive row
// fragments (16 planar vectors each) g0 .. g0 + rf - 1 of the plan, rf = 4 (up to 20 column fragments, e.g. one row tile of the plan)
// or 5 (up to 16 column fragments): 20 accumulator fragments per wave either way.  nbf = active 8-kz column fragments per row
// fragment, rf x 8 bit.
struct SkItem { int g0, rf, ct, c0, c1; unsigned long long nbf; };
// parameter block of sk_gemm's projecting epilogue (device memory): weights [R_pad][C_pad], z-class phases class-major [nzc][C_pad]
struct SkProj { const double *wfull, *tzt; int nzc, cpad; };
// sk_gemm's work list, one fixed-size row per workgroup: the segments it runs, each with its output slots -- one per row tile of the
// plan the band touches (sga: row tile g0 >> 2, sgb: the next one or -1), numbered tile-major (SkTile::item0), for the partial-tile
// mode; sg = the segment's own index, the slot of its band-local piece in the projecting mode -- and the first entry's nseg = how
// many are used.  One load replaces the seg_ptr -> seg_idx -> items chain at the top of every workgroup.
struct SkWItem { int g0, rf, ct, c0, c1, sg, sga, sgb, nseg, slab_slot /*SkFuse: where this segment's sum of q z goes, or -1*/; unsigned long long nbf; };
struct SkTile { int rt, ct, nba, item0, nsplit; unsigned nbf; };     // one (row tile, col tile) of the plan: its output slots are item0 .. item0+nsplit-1

struct RealParams {           // real-space pair kernels
  double g_ewald, eta, cut_coulsq;    // cut_coulsq already min(cut_coul^2, (5.8/g)^2)  fix_conp.cpp:1237-1240
  int ntypes;
  const double *cutsq;                // [(ntypes+1)^2]
  int ehgo;                           // EHGO pair mode (fix_conp.cpp:1561-1573): per type-pair eta_ij, fo_ij
  const double *eta_ij, *fo_ij;       // [(ntypes+1)^2]
  const double *u0_i;                 // [ntypes+1] (post-force self energy :1182-1199)
};

// everything one electrode row of b needs besides the k-space partials' origin (b_real_combine_kernel; the fused tail of
// b_zc_dot_kernel): real-space rows [row0, row1), the four k-space partial slots, the slab term
struct BRowArgs {
  int ne, ne_pad, row0, row1;
  const int *row_ptr, *ele_atom, *oth_atom;
  const double *x, *q;
  const int *type;
  RealParams rp;
  int add_k;
  const double *bk;
  int slab;
  const double *ele_z, *slab_part;
  int n_slab_part;
  double slab_pref;
  double *b_out, *slab_out;
  const double *breal;      // NULL: the real-space pair sums are formed by whoever assembles the row; else they are read here
};
inline BRowArgs make_brow(int ne, int ne_pad, int row0, int row1, const int *row_ptr, const int *ele_atom, const int *oth_atom,
                          const double *x, const double *q, const int *type, RealParams rp, int add_k, const double *bk, int slab,
                          const double *ele_z, const double *slab_part, int n_slab_part, double slab_pref, double *b_out,
                          double *slab_out) {
  BRowArgs a;
  a.ne = ne; a.ne_pad = ne_pad; a.row0 = row0; a.row1 = row1; a.row_ptr = row_ptr; a.ele_atom = ele_atom; a.oth_atom = oth_atom;
  a.x = x; a.q = q; a.type = type; a.rp = rp; a.add_k = add_k; a.bk = bk; a.slab = slab; a.ele_z = ele_z; a.slab_part = slab_part;
  a.n_slab_part = n_slab_part; a.slab_pref = slab_pref; a.b_out = b_out; a.slab_out = slab_out; a.breal = nullptr;
  return a;
}

// Small systems (round 4): the phase tables of a segment's atoms are computed by the segment's own workgroup in front of its chunk
// loop (sk_phase_prologue: the arithmetic of elyte_phase_kernel; several bands compute the same atoms' entries -- identical values
// to identical addresses), the real-space pair sums ride in spare workgroups of the same launch: no elyte_phase launch at all.
struct SkFuse {
  int on;                         // 0: the tables were filled by elyte_phase_kernel
  int nl, kzt, nwg_sk;            // charged electrolyte atoms; kz values per column tile; workgroups that run segments (the rest: pair rows)
  const int *elyte_idx;
  const double *x, *q;
  double ux, uy, uz;
  double2 *Xt, *Yt, *Zs;
  double *qc, *slab_part;         // slab_part[SkWItem::slab_slot]: sum of q z over the segment's atoms (segments of band 0 of column tile 0)
  BRowArgs rows;
  double *breal_out;
};

struct PppmDev {              // device view of PppmPlan
  int nx, ny, nz, order, nlower, nfft;
  double shift, shiftone, delinv[3], delvolinv, boxlo[3];
  const double *rho_coeff, *greensfn, *twid[3];
};
// pppm_conp.cpp:269-316 on the device (conp_pppm.hip): bk slot 0 <- PPPM k-space b of all electrode atoms
void launch_pppm_b(hipStream_t s, const PppmDev &pd, int nl, const int *elyte_idx, const double *x, const double *q, int ne,
                   int ne_pad, const int *egrid, const double *ew, double *re, double *im, double *slab_part, int *n_slab_part,
                   double *bk, bool *im_clean /* in/out: `im` is all zero -- then it receives the charges and no clearing launch is needed */,
                   const BRowArgs *pairs = nullptr /*with breal_out: the real-space pair sums of these rows ride in the spread launch*/,
                   double *breal_out = nullptr, BRowArgs *fin = nullptr /*the gather completes the rows of b (needs fin->breal)*/,
                   double *keep_rho = nullptr /*[nfft]: receives the electrolyte density brick of this update*/);

// PPPM coupling beyond b (pppm_conp.cpp:385-534) and the pair part of compute potential/atom (compute_potential_atom.cpp:223-308)
void launch_pppm_density(hipStream_t s, const PppmDev &pd, int n, const int *idx, const double *x, const double *q, double *rho,
                         double *slab_scratch /*[>= 1025]*/);
void launch_pppm_poisson(hipStream_t s, const PppmDev &pd, double *re /*rho in, u_brick out*/, double *im);
void launch_pppm_probe(hipStream_t s, const PppmDev &pd, int n, const int *idx, const double *x, const double *q, const double *u,
                       double self, double *out /*indexed by atom*/);
void launch_potential_pair(hipStream_t s, int inum, const int *ilist, const int *numneigh, const int *first, const int *neigh,
                           int nlocal, int newton, const double *x, const double *q, const int *type, const int *sel,
                           const int *etasel, int ntypes, const double *cutsq, double cut_coulsq, double g_ewald, double eta,
                           double *potential);

// ---- the z-window form of the structure-factor contraction (conp_zn.hip, round 5) ------------------------------------------------
// item = (row tile: 64 planar vectors, chunk range [c0, c1) of the z-ordered electrolyte list, window origin g0 on the grid, slot of
// its piece [class][128 rows] in the pieces buffer)
void launch_zn_ptable(hipStream_t s, const DevPlan &pl, int kzt, int nzc, int n, const double *tzt /*[nzc][C_pad]*/,
                      const double *phihat /*[nz]*/, const double2 *cs /*[n]: (cos, sin)(2 pi k / n)*/, double *P /*[R_pad][nzc][n]*/);
// rough electrodes: the ranges' raw windows (launch_zn_gemm with P == nullptr) -> rows on the z grid -> G, w o G (sk_reduce's outputs)
void launch_zn_dtable(hipStream_t s, const DevPlan &pl, int kzt, int n, const double *phihat, const double2 *cs, double *Dt /*[n][C_pad]*/);
void launch_zn_windows_to_g(hipStream_t s, const DevPlan &pl, int kzt, int n, int n_own, const int *own_rt, int nrg, int ncol, const int *cov_ptr,
                            const int2 *cov_ent, const double *raw, double *grid, const double *Dt, double *G, double *Gwf);
void launch_zn_gemm(hipStream_t s, const DevPlan &pl, int ncf /*2 or 3 column fragments*/, const ZnItem *items, int nitems, const double2 *Xt,
                    const double2 *Yt, const double *Bt, const double *P, int n, int nzc, double *pieces, int piece_stride);

// ---- per-step electrolyte path -----------------------------------------------------------------
void launch_ghost_fill(hipStream_t s, int nlocal, int nghost, const int *owner, const int *img, double px, double py, double pz,
                       double *x, double *q);
void launch_elyte_phase(hipStream_t s, int nl, int nl_pad, const int *elyte_idx, const double *x, const double *q,
                        double ux, double uy, double uz, int kxmax, int kymax, int nz, int kzt, int nrz, double2 *Xt,
                        double2 *Yt, double2 *Zs, double *qc, double *slab_part, int *n_slab_part,
                        const BRowArgs *rows /*NULL, or: the real-space pair sums of these rows ride along, into breal_out*/,
                        double *breal_out, int j0 /*tables for the atoms [j0, j1) of the compact list only (a rank's share)*/, int j1,
                        const ZnWindow *zw = nullptr /*the z-window form: the window matrix instead of the z phase seeds*/);
bool zc_final_fits(int n_own, int nzc);
void launch_sk_gemm(hipStream_t s, const DevPlan &pl, const SkWItem *witems /*[nwg][maxseg]*/, int maxseg, int nwg, int nl_pad,
                    const double2 *Xt, const double2 *Yt, const double2 *Zs, const double *qc, double *part, const SkProj *proj = nullptr,
                    const SkFuse *fuse = nullptr /*small systems: phase tables and pair sums inside this launch (DEVICE copy of the block)*/,
                    int fuse_rows = 0 /*its rows.ne*/, unsigned *ticket = nullptr /*zeroed by workgroup 0: b_zc_fused_kernel's hand-off word*/);
int sk_hc_stride();           // doubles per segment of sk_gemm's projected output
int sk_hc_max_classes();      // most z classes the projecting mode takes
void launch_project_zclass_pieces(hipStream_t s, const DevPlan &pl, int ne_pad, int n_own, const int *own_rt, int nzc, const double *Hp,
                                  const int *slot_ptr, const int *slot_idx, bool presum /*hc_sum first: many pieces, or bands that are
                                  not row tiles of the plan*/, const int *frag_ptr /*[nfrag + 1]*/, const int2 *frag_ents /*per row fragment of the plan: its pieces
                                  (offset of the fragment's first 'a' row, the band's rf)*/, int nfrag, const double *Rp, const double2 *Xe,
                                  const double2 *Ye, const int *own_pv, const int *zclass, double *Hc, double *bk_part, const BRowArgs *fin,
                                  const BRowArgs *pairs = nullptr /*with breal_out: the pair sums ride in hc_sum's launch*/,
                                  double *breal_out = nullptr,
                                  unsigned *ticket = nullptr /*with fin and pairs: ONE launch, the pieces' sums handed over inside it (round 5)*/,
                                  unsigned spin_limit = 1u << 16,
                                  bool wide = false /*many pieces per fragment: 32 threads per element in the sum*/);
void launch_sk_reduce(hipStream_t s, const DevPlan &pl, const SkTile *tiles, int ntiles, int max_nsplit, double *part, double *G,
                      double *Gwf);
void launch_sfac_gather(hipStream_t s, int kcount, int C_pad, int PT, const int *sf_row_a, const int *sf_col_c,
                        const int *k_sign, const double *G, double *sfacrl, double *sfacim);
void launch_b_project(hipStream_t s, const DevPlan &pl, int ne_pad, const int *ct_ptr /*[n_col_tiles+1]*/, const SkTile *tiles,
                      const double *Gwf, const double *Rp, const double *Tz, double *bk_part /*[4][ne_pad] overwritten*/);
// planar-electrode fast path of the projection (<= 64 distinct electrode z values)
void launch_reduce_project_zclass(hipStream_t s, const DevPlan &pl, const SkTile *tiles, int ntiles, int max_nsplit, double *part,
                                  double *G, int ne_pad, int n_own, const int *own_rt, int nzc, const double *Tzc, const double *Rp,
                                  const double2 *Xe /*[kxmax+2][ne_pad] electrode axis phases, last row zero*/, const double2 *Ye /*[kymax+1][ne_pad]*/,
                                  const int *own_pv /*[n_own][64] packed planar vectors of the own row tiles*/, const int *zclass, double *Hc, double *bk_part,
                                  const BRowArgs *fin /*non-NULL (needs fin->breal): the dot kernel finishes b itself*/);
void launch_b_project_zclass(hipStream_t s, const DevPlan &pl, int ne_pad, const int *rt_mine, int n_own, const int *own_rt, int nzc, const double *Gwf,
                             const double *Tzc /*[C_pad][64]*/, const double *Rp, const double2 *Xe, const double2 *Ye,
                             const int *own_pv, const int *zclass /*[ne_pad]*/,
                             double *Hc /*[4][R_pad][64]*/, double *bk_part /*[4][ne_pad]*/, const BRowArgs *fin);
// this rank's contribution to b in one launch: k-space halves + slab (rank 0) + real-space rows row0..row1
void launch_b_real_combine(hipStream_t s, int ne, int ne_pad, int row0, int row1, const int *row_ptr, const int *ele_atom,
                           const int *oth_atom, const double *x, const double *q, const int *type, RealParams rp, int add_k,
                           const double *bk, int slab, const double *ele_z, const double *slab_part, int n_slab_part,
                           double slab_pref, double *b_out, double *slab_out,
                           const double *breal = nullptr /*pair sums formed earlier in this update, or NULL: formed here*/);
// the same GEMV + charge write with the matrix taken as symmetric: packed lower-triangle tiles, half the bytes (conp_kernels.hip)
size_t sym_packed_doubles(int ne_pad);
// stat [2] (device): receives the bit patterns of max |S_ij| and max |S_ij - S_ji| -- how symmetric the matrix is
void launch_sym_pack(hipStream_t s, int ne, int ne_pad, const double *S, double *Spk, unsigned long long *stat);
void launch_sym_gemv_finish(hipStream_t s, int n, int ne_pad, const double *Spk, const double *b, double *yp /*[ne_pad / 128][ne_pad]*/,
                            double *y, const double *elesetq, const double *eleinitq, double potdiff, const int *atoms_ptr,
                            const int *atoms_of, const int *atoms_row /*row of every CSR entry*/, double *q_ele, double *q_atoms);
void launch_gemv_rows(hipStream_t s, int n, int row0, int row1, const double *S, const double *b, double *y);
// all rows + the charge write of plain `fix conp` in one launch (atoms_ptr / atoms_of: electrode row -> its owned and ghost atoms)
void launch_gemv_finish(hipStream_t s, int n, const double *S, const double *b, double *y, const double *elesetq,
                        const double *eleinitq, double potdiff, const int *atoms_ptr, const int *atoms_of, double *q_ele,
                        double *q_atoms);
void launch_charge_finish(hipStream_t s, int ne, int nall, const int *atom2eleall, const int *elecheck, const double *eleallq,
                          const double *elesetq, const double *eleinitq, double potdiff, const double *d_potdiff, double *q_ele,
                          double *q_atoms, double *left_out);
void launch_cond_potdiff(hipStream_t s, int ne, const double *setzvec, const double *eleallq, const double *slab_part,
                         int n_slab_part, double lz, double rightcharge, double vmult, double *out);
void launch_conq_potdiff(hipStream_t s, const double *left, double rightcharge, double totsetq, int one_electrode, double *out);
size_t b_rows_scratch_bytes(int ne, size_t nneigh);
void launch_atom2eleall(hipStream_t s, int nall, int npairs, const int *pairs /*[npairs][2] = (atom, eleall)*/, int *atom2eleall /*[nall]*/);
void launch_build_b_rows(hipStream_t s, int inum, size_t nneigh, const int *ilist, const int *numneigh, const int *first,
                         const int *neigh, const int *arow, int nlocal, int newton, int ne, void *scratch, size_t scratch_bytes,
                         int *row_ptr, int *ele, int *oth, unsigned *np_pinned /*page-locked host word: the number of pairs, valid
                         after the stream has been synchronised*/);
void launch_post_force(hipStream_t s, int inum, const int *ilist, const int *numneigh, const int *first, const int *neigh, int nlocal,
                       int nall, int newton, const double *x,
                       const double *q, const int *type, const int *atom2eleall, RealParams rp, double qqrd2e, double *f,
                       double *acc /*[9]: eng_coul, virial[6], sum q^2 of owned electrode atoms, contributing pairs*/, bool clear_f);
void launch_left_sum(hipStream_t s, int ne, const int *elecheck, const double *v, double *out);
void launch_results_out(hipStream_t s, int ne, const int *elecheck, const double *v, double *scal, bool do_left, const double *qele,
                        double *host_q /*page-locked host memory*/, double *host_scal);

// ---- once-per-run matrix work ------------------------------------------------------------------
// electrode phase tables on the device (conp_tables.hip; km_ewald.cpp:426-531).  seeds: [6][ne] = (cos, sin)(unitk_c x_ic) per axis,
// from the host's libm; the target buffers must be zeroed (padding rows and atoms stay zero).
void launch_ele_tables(hipStream_t s, const DevPlan &pl, int kzt, int ne, int ne_pad, const double *seeds, double2 *Xe /*[kxmax+2][ne_pad]*/,
                       double2 *Ye /*[kymax+1][ne_pad]*/, double *Tz /*[C_pad][ne_pad]*/, double *Rp /*[R_pad][ne_pad]*/);
// Tzc[t][c] = Tz[t][rep[c]] ([C_pad][64]) and its class-major copy TzcT ([nzc][C_pad])
void launch_ele_zclass(hipStream_t s, int C_pad, int ne_pad, int nzc, const int *rep, const double *Tz, double *Tzc, double *TzcT);
int a_kspace_nsplit(int ne_pad, int num_cus, int nchunk, int nranks);
// planar electrodes: the same matrix through the z-class factorisation (contraction over the planar rows only)
void launch_a_kspace_zclass(hipStream_t s, const DevPlan &pl, int ne, int ne_pad, int nzc, const double *Rp, const double *Tzc,
                            const int *zclass, double *Wz /*[R_pad * nzc * nzc] scratch*/, double *A, int rank, int nranks);
// rank r of nranks computes the lower-triangle tiles r, r + nranks, ... into its zero-initialised A
void launch_a_kspace(hipStream_t s, const DevPlan &pl, int ne, int ne_pad, const double *Rp, const double *Tz, double *A, int nsplit,
                     const int *chunk_group, int rank, int nranks);
void launch_a_diag_slab(hipStream_t s, int ne, double diag_k, double diag_self, const double *diag_self_atom /*[ne] or NULL*/,
                        int slab, double pref, const double *ele_z, double *A);
void launch_a_real(hipStream_t s, int ne, int row0, int row1, const int *row_ptr, const int *ele_atom, const int *oth_atom,
                   const int *col, const double *x, const int *type, RealParams rp, double *A);
void launch_a_symmetrise(hipStream_t s, int ne, double *A);
void launch_inv_project(hipStream_t s, int n, double *A, int use_mask, const unsigned char *mask, double *ainve,
                        double *totinve /*device scalar*/, int apply);
void launch_inv_project_apply(hipStream_t s, int n, double *A, const double *ainve, const double *totinve);
// in-place inverse (conp_inverse.hip): blocked Gauss-Jordan with partial pivoting; *info != 0 -> singular
size_t inverse_workspace_doubles(int n);
// returns true when the multi-workgroup panel was used (then info == -7 means "a grid barrier timed out": restore M, repeat
// with multi_wg = false)
// symmetric positive definite matrices: no pivot search, no grid barrier (conp_inverse.hip); info = -8: not positive definite
void launch_symmetry_check(hipStream_t s, int n, const double *M, int *flag_dev);
void launch_inverse_spd(hipStream_t s, int n, double *M, double *work, int *info /*[1]*/);
bool launch_inverse(hipStream_t s, int n, double *M, double *work, int *piv_all /*[n]*/, int *info /*[1]*/, int num_cus,
                    bool multi_wg, int max_wg /*0: no cap*/, unsigned spin_limit /*polls of the panel's grid barrier before info = -7*/);
// CG (fix_conp.cpp:864-930): state vectors on device; returns via *d_done
void launch_cg_init(hipStream_t s, int n, const double *A, const double *b, double *q, double *res, double *p,
                    double *scal /*[8]*/);
void launch_cg_iter(hipStream_t s, int n, const double *A, double *q, double *res, double *p, double *ap, double *scal,
                    double tolerance, int *done, int iter, double *hist);
// one launch per CG iteration (update of iteration iter - 1 repeated by every workgroup + its rows of the matvec); see the kernel
// round 5: the whole CG solve as one persistent launch (n <= 4096); false: not all workgroups can be resident
bool cg_persist_fits(int n);
bool launch_cg_persist(hipStream_t s, int num_cus, int n, const double *A, const double *b, double *q, double *ap2, double *scal,
                       double tolerance, int maxiter, double *hist, double *host_ctl, unsigned *ticket, unsigned *ticket_next,
                       unsigned spin_limit);
bool cg_step_fits(int n);
void launch_cg_step(hipStream_t s, int n, const double *A, const double *b, double *q, double *res2 /*[2][n]*/, double *p2 /*[2][n]*/,
                    double *ap2 /*[2][n]*/, double *scal, double tolerance, int *done, int iter, double *hist, int mode,
                    double *host_ctl = nullptr /*page-locked host memory: mode 4 stores n_ctl doubles of scal there*/, int n_ctl = 0);

}  // namespace conp
