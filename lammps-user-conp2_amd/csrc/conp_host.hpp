// Host-side (CPU, integer / setup) pieces of the MI355X constant-potential solver:
//   KTables   : the Ewald k-vector tables in the reference's order (bit-exact index contract)
//   KPlan     : the (planar p, kz m) factorisation of those tables that the HIP kernels work on
//   EleIndex  : electrode index bookkeeping of FixConp::post_neighbor
//   PairRows  : LAMMPS half lists -> electrode-row CSR for the real-space kernels
// Pure C++17, no HIP.  Reference citations are file:line in /root/reference.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

namespace conp {

constexpr int NEIGHMASK = 0x3FFFFFFF;  // LAMMPS lmptype.h (SBBITS 30); used at fix_conp.cpp:1253,1324

// ------------------------------------------------------------------------------------------------
// KTables: km_ewald.cpp:63-132 (conp_setup), :277-283 (rms), :285-364 (make_kvecs_ewald),
//          :366-381 (make_ug_from_kvecs), :383-424 (make_kxy_list_from_kvecs)
// ------------------------------------------------------------------------------------------------
struct KTables {
  double g_ewald = 0, accuracy = 0, slab_volfactor = 1, volume = 0, gsqmx = 0, ug_tot = 0;
  double unitk[3] = {0, 0, 0};
  int slabflag = 0;
  int kxmax = 0, kymax = 0, kzmax = 0, kmax = 0, kmax3d = 0;
  int kcount = 0, kcount_flat = 0, kcount_expand = 0;
  int kcount_dims[7] = {0, 0, 0, 0, 0, 0, 0};
  std::vector<int> kxvecs, kyvecs, kzvecs, kxy_list, kz_list;
  std::vector<double> ug;

  void build(double g_ewald, double accuracy_abs, double slab_volfactor, int slabflag, double xprd, double yprd,
             double zprd, double qsqsum, int64_t natoms, double qqrd2e, double dielectric);
};

// ------------------------------------------------------------------------------------------------
// KPlan: every half-space k of the reference list is (planar vector p = (kx, +-ky), kz index m >= 0, sign of kz).
// The structure factors of all of them follow from four real matrix products over the electrolyte atoms j
//     G[(p,a|b)][(m,c|s)] = sum_j  q_j {cos,sin}(theta_pj) * {cos,sin}(m uz z_j)
// (S_re(p,+-m) = CC -+ SS, S_im(p,+-m) = SC +- CS), and the k-space b vector / A matrix are bilinear forms in G
// weighted by w(p,m) = sum over the signs present of 2 ug.  Rows/cols are laid out for the MFMA tiles.
// ------------------------------------------------------------------------------------------------
struct KPlan {
  static constexpr int PT = 64;       // planar vectors per row tile (128 G rows: 64 'a' rows then 64 'b' rows)
  static constexpr int CT_BLK = 10;   // 16-kz blocks of column SPACE per column tile (320 G columns); a tile USES the first kzt / 16 of them
  static constexpr int CT_COLS = 32 * CT_BLK;

  int kxmax = 0, kymax = 0, nz = 0;   // nz = kzmax + 1 (m = 0 .. kzmax)
  int np = 0;                         // planar vectors, the origin first: singles + the smallest pairs (whole tiles), then the (+ky, -ky) pairs by |k_p|^2
  int paired_lo = 0, paired_hi = 0;   // the row tiles [paired_lo, paired_hi) hold 32 whole pairs each, (+, -) on (even, odd) indices
  std::vector<int> p_ikx, p_iky, p_sgn;   // per p: |kx|, |ky|, sign of ky (+1/-1); origin = (0,0,+1)
  std::vector<int> flat2p;                // reference flat index (x axis, y axis, (k,+-l,0)) -> p ; z-axis entries -> -1
  std::vector<int> k_p, k_m, k_sign;      // per reference k index
  int nblk = 0;                           // ceil(nz / 16)
  int kzt = 16 * CT_BLK;                  // kz values per column tile: nz spread evenly over the tiles (a multiple of 8, <= 160), so that
                                          // no tile is a runt (slab geometry, nz = 378: 3 x 128 instead of 160 + 160 + 58)
  int n_row_tiles = 0, n_col_tiles = 0, R_pad = 0, C_pad = 0;
  std::vector<int> nba_rc;                // [n_col_tiles][n_row_tiles]: leading 16-kz blocks OF THE COLUMN TILE that hold a listed k (sphere cut)
  std::vector<int> nfa_fc;                // [n_col_tiles][n_row_tiles * 4]: the same per 16 planar vectors (one MFMA row fragment), in
                                          // 8-kz COLUMN FRAGMENTS of the tile
  std::vector<double> w;                  // [np][nz] weights
  std::vector<double> wfull;              // [R_pad][C_pad] weights expanded to G's layout (0 in padding)
  std::vector<int> sf_row_a, sf_col_c;    // per reference k: G row of (p,'a') and col of (m,'c') (b row = +PT, s col = +8)

  // G column layout: one 16-column MFMA fragment per 8 kz values: 8 'c' (cos) columns then 8 's' (sin) columns, so that the sphere
  // cut is applied in steps of 8 kz per row fragment.  Two fragments = the 32 columns of one 16-kz block: everything that works
  // per block (nb_act, the reductions, the projections, the SYRK's chunks) sees the same column set as with 16 + 16.
  int row_a(int p) const { return (p / PT) * (2 * PT) + (p % PT); }
  int row_b(int p) const { return row_a(p) + PT; }
  int col_c(int m) const { const int ct = m / kzt, ml = m - ct * kzt; return CT_COLS * ct + 16 * (ml >> 3) + (ml & 7); }
  int col_s(int m) const { return col_c(m) + 8; }
  int nba(int rt, int ct) const { return nba_rc[(size_t)ct * n_row_tiles + rt]; }
  // active column fragments of row fragment f (planar vectors 16 f .. 16 f + 15 of row tile rt) inside col tile ct, packed 4 x 8 bit
  unsigned nfa16(int rt, int ct) const {
    unsigned v = 0;
    for (int f = 0; f < 4; ++f) v |= (unsigned)nfa_fc[((size_t)ct * n_row_tiles + rt) * 4 + f] << (8 * f);
    return v;
  }

  void build(const KTables &kt);
};

// ------------------------------------------------------------------------------------------------
// PppmPlan: what PPPMCONP inherits from LAMMPS PPPM for the b vector (pppm_conp.cpp:126-344): mesh geometry, the
// charge-assignment polynomial coefficients (PPPM::compute_rho_coeff) and the ik influence function
// (PPPM::compute_gf_ik, compute_gf_denom).  Those LAMMPS routines are not part of the reference repository; they are
// restated from the Hockney-Eastwood P3M formulation as LAMMPS implements it (parity unpinned).
// ------------------------------------------------------------------------------------------------
struct PppmPlan {
  static constexpr int OFFSET = 16384;   // pppm_conp.cpp:32
  static constexpr int MAXORDER = 8;
  int nx = 0, ny = 0, nz = 0, order = 0, nlower = 0, nupper = 0, nfft = 0;
  double shift = 0, shiftone = 0, delinv[3] = {0, 0, 0}, delvolinv = 0, boxlo[3] = {0, 0, 0}, volume = 0;
  std::vector<double> rho_coeff;   // [order][order]: coefficient l of stencil point m
  std::vector<double> greensfn;    // [nz][ny][nx]
  std::vector<double> twid[3];     // per axis: cos, sin (2 pi t / n) interleaved

  void build(int nx, int ny, int nz, int order, double g_ewald, double slab_volfactor, const double *boxlo, const double *prd);
  void rho1d(double dx, double *w) const;   // PPPM::compute_rho1d, one axis
};

// electrode phase tables: the host's share is the 3 Ne seed pairs (cos, sin)(unitk_c x_ic) of km_ewald.cpp:440-442, [6][ne] = cx, sx,
// cy, sy, cz, sz; the recurrences and products of sincos_a_ele (:443-477) run on the device (conp_tables.hip)
void electrode_seeds(const KTables &kt, int ne, const double *xele /*[ne][3]*/, std::vector<double> &seeds);

// ------------------------------------------------------------------------------------------------
// RankOps: the collectives FixConp::post_neighbor / linalg_init make on `world` (fix_conp.cpp:415, 492, 523, 535).  The default
// is one rank; conp_fix.cpp implements it on the host's conp_comm callbacks for spatially decomposed runs.
// ------------------------------------------------------------------------------------------------
struct RankOps {
  virtual ~RankOps() {}
  virtual int nranks() const { return 1; }
  virtual int rank() const { return 0; }
  virtual void allreduce_max_int(int *, int) {}
  virtual void allgather_int(int v, int *out) { out[0] = v; }
  virtual void allgatherv_int(const int *send, int n, int *recv, const int * /*counts*/, const int * /*displs*/) {
    for (int i = 0; i < n; ++i) recv[i] = send[i];
  }
};

// ------------------------------------------------------------------------------------------------
// EleIndex: FixConp::post_neighbor (fix_conp.cpp:468-539) + linalg_init's tag2eleall sizing (:413-416).
// ------------------------------------------------------------------------------------------------
struct EleIndex {
  int elenum = 0, elenum_all = 0, elytenum = 0, maxtag_all = -1;
  std::vector<int> ele2tag, ele2eleall, tag2eleall, eleall2tag, eleall2ele, elecheck_eleall, elebuf2eleall;
  std::vector<int> elenum_list, displs;   // per rank: owned electrode atoms and their offsets in gathered buffers (:492-506)
  std::vector<int> tag2local;   // atom->map(tag) for owned atoms
  std::vector<int> ele_local;   // local indices of the owned electrode atoms, ascending (scratch of post_neighbor)
  bool initialised = false;

  void linalg_init(int nlocal, const int *tag, RankOps *ops = nullptr);
  // returns true when elenum_all grew (the reference then reallocates A, b, q ... :510-525)
  bool post_neighbor(int nlocal, const int *tag, const int *echeck, bool *elyte_grew, RankOps *ops = nullptr);
  void map_atoms(int nlocal, const int *tag);
  // FixConp::a_read (fix_conp.cpp:753-772): the matrix file's tag row becomes the permanent numbering
  void renumber_from_tags(const std::vector<int> &file_tags, int nlocal, const int *tag, const int *echeck, RankOps *ops = nullptr);
};

// ------------------------------------------------------------------------------------------------
// PairRows: pairs of a LAMMPS half list regrouped by electrode row (global eleall index), keeping list order
// inside each row.  Restates the membership logic of blist_coul_cal (fix_conp.cpp:1313-1353) and
// alist_coul_cal (:1242-1276); the distance tests stay in the kernel because x changes every step.
// ------------------------------------------------------------------------------------------------
struct PairRows {
  std::vector<int> row_ptr;   // [Ne+1]
  std::vector<int> ele_atom;  // per pair: atom index of the electrode member (owned or ghost)
  std::vector<int> oth_atom;  // per pair: atom index of the partner (b: electrolyte atom; a: second electrode atom)
  std::vector<int> col;       // a-list only: eleall index of the partner
  int64_t npairs() const { return (int64_t)ele_atom.size(); }
};

struct ListView {
  int inum = 0;
  const int *ilist = nullptr, *numneigh = nullptr, *first = nullptr, *neigh = nullptr;
};

void build_b_rows(const ListView &l, int nlocal, const int *tag, const int *echeck, const EleIndex &idx, bool newton,
                  PairRows &out);
// every listed pair with exactly one electrode member, as (owner i, neighbour j): the pair set of
// blist_coul_cal_post_force (fix_conp.cpp:1411), which has no newton/ghost filter
void build_pf_pairs(const ListView &l, const int *echeck, std::vector<int> &pi, std::vector<int> &pj);
void build_a_rows(const ListView &l, int nlocal, const int *tag, const int *echeck, const EleIndex &idx, bool newton,
                  PairRows &out);

}  // namespace conp
