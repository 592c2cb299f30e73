// Host-side setup and bookkeeping.  See conp_host.hpp.
#include "conp_host.hpp"

#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

namespace conp {

static constexpr double MY_PI = 3.14159265358979323846;  // LAMMPS math_const.h

// ------------------------------------------------------------------------------------------------
// KTables
// ------------------------------------------------------------------------------------------------
namespace {
// RMS force-error estimate of an axis truncated at km (km_ewald.cpp:277-283)
double axis_rms(double g, int km, double prd, int64_t natoms, double q2) {
  return 2.0 * q2 * g / prd * std::sqrt(1.0 / (MY_PI * km * natoms)) *
         std::exp(-MY_PI * MY_PI * km * km / (g * g * prd * prd));
}
}  // namespace

void KTables::build(double g, double acc, double slabf, int slab, double xprd, double yprd, double zprd,
                    double qsqsum, int64_t natoms, double qqrd2e, double dielectric) {
  g_ewald = g; accuracy = acc; slab_volfactor = slabf; slabflag = slab;
  const double q2 = qsqsum * qqrd2e / dielectric;       // :79
  const double zprd_slab = zprd * slab_volfactor;       // :84
  volume = xprd * yprd * zprd_slab;
  unitk[0] = 2.0 * MY_PI / xprd;
  unitk[1] = 2.0 * MY_PI / yprd;
  unitk[2] = 2.0 * MY_PI / zprd_slab;
  const double prds[3] = {xprd, yprd, zprd_slab};
  int kmaxes[3];
  for (int c = 0; c < 3; ++c) {                         // :93-113
    int km = 1;
    while (axis_rms(g, km, prds[c], natoms, q2) > acc) ++km;
    kmaxes[c] = km;
  }
  kxmax = kmaxes[0]; kymax = kmaxes[1]; kzmax = kmaxes[2];
  kmax = std::max(kxmax, std::max(kymax, kzmax));
  kmax3d = 4 * kmax * kmax * kmax + 6 * kmax * kmax + 3 * kmax;
  const double gx = unitk[0] * unitk[0] * kxmax * kxmax;  // :120-126
  const double gy = unitk[1] * unitk[1] * kymax * kymax;
  const double gz = unitk[2] * unitk[2] * kzmax * kzmax;
  gsqmx = std::max(gx, std::max(gy, gz));
  gsqmx *= 1.00001;

  double usq[3];
  for (int c = 0; c < 3; ++c) usq[c] = unitk[c] * unitk[c];
  kxvecs.clear(); kyvecs.clear(); kzvecs.clear();
  for (int &d : kcount_dims) d = 0;
  auto push = [&](int a, int b, int c) { kxvecs.push_back(a); kyvecs.push_back(b); kzvecs.push_back(c); };
  // axes :297-310
  for (int c = 0; c < 3; ++c)
    for (int m = 1; m <= kmaxes[c]; ++m) {
      double sqk = m * m * usq[c];
      if (sqk <= gsqmx) { push(c == 0 ? m : 0, c == 1 ? m : 0, c == 2 ? m : 0); ++kcount_dims[c]; }
    }
  // coordinate planes :318-342 : (k,+-l,0) (0,k,+-l) (k,0,+-l)
  const int planeA[3] = {0, 1, 0}, planeB[3] = {1, 2, 2};
  for (int pl = 0; pl < 3; ++pl) {
    const int ca = planeA[pl], cb = planeB[pl];
    for (int k = 1; k <= kmaxes[ca]; ++k)
      for (int l = 1; l <= kmaxes[cb]; ++l) {
        double sqk = k * k * usq[ca] + l * l * usq[cb];
        if (sqk <= gsqmx) {
          for (int sg = 1; sg >= -1; sg -= 2) {
            int v[3] = {0, 0, 0};
            v[ca] = k; v[cb] = sg * l;
            push(v[0], v[1], v[2]);
          }
          ++kcount_dims[3 + pl];
        }
      }
  }
  // bulk :346-359 : (k,l,m) (k,l,-m) (k,-l,m) (k,-l,-m)
  for (int k = 1; k <= kxmax; ++k)
    for (int l = 1; l <= kymax; ++l)
      for (int m = 1; m <= kzmax; ++m) {
        double sqk = k * k * usq[0] + l * l * usq[1] + m * m * usq[2];
        if (sqk <= gsqmx) {
          push(k, l, m); push(k, l, -m); push(k, -l, m); push(k, -l, -m);
          ++kcount_dims[6];
        }
      }
  kcount = (int)kxvecs.size();
  kcount_flat = kcount_dims[0] + kcount_dims[1] + kcount_dims[2] + 2 * kcount_dims[3];
  kcount_expand = kcount_dims[4] + kcount_dims[5] + 2 * kcount_dims[6];

  // ug :366-381
  ug.assign(kcount, 0.0);
  const double g_ewald_sq_inv = 1.0 / (g_ewald * g_ewald);
  const double preu = 4.0 * MY_PI / volume;
  ug_tot = 0;
  for (int k = 0; k < kcount; ++k) {
    double sqk = kxvecs[k] * kxvecs[k] * unitk[0] * unitk[0];
    sqk += kyvecs[k] * kyvecs[k] * unitk[1] * unitk[1];
    sqk += kzvecs[k] * kzvecs[k] * unitk[2] * unitk[2];
    ug[k] = preu * std::exp(-0.25 * sqk * g_ewald_sq_inv) / sqk;
    ug_tot += 2 * ug[k];
  }

  // gather lists :383-424
  kxy_list.assign(kcount_expand, 0);
  kz_list.assign(kcount_expand, 0);
  const int zbase = kcount_dims[0] + kcount_dims[1] - 1;
  int kf = kcount_flat, e = 0;
  for (int n = 0; n < kcount_dims[4]; ++n, ++e, kf += 2) {
    kxy_list[e] = kyvecs[kf] + kcount_dims[0] - 1;
    kz_list[e] = kzvecs[kf] + zbase;
  }
  for (int n = 0; n < kcount_dims[5]; ++n, ++e, kf += 2) {
    kxy_list[e] = kxvecs[kf] - 1;
    kz_list[e] = kzvecs[kf] + zbase;
  }
  int kxy = kcount_dims[0] + kcount_dims[1] + kcount_dims[2];
  for (int n = 0; n < kcount_dims[6]; ++n, e += 2, kf += 4) {
    while (kxvecs[kxy] != kxvecs[kf] || kyvecs[kxy] != kyvecs[kf]) kxy += 2;
    kxy_list[e] = kxy;
    kxy_list[e + 1] = kxy + 1;
    kz_list[e] = kz_list[e + 1] = kzvecs[kf] + zbase;
  }
}

// ------------------------------------------------------------------------------------------------
// KPlan
// ------------------------------------------------------------------------------------------------
void KPlan::build(const KTables &kt) {
  kxmax = kt.kcount_dims[0];
  kymax = kt.kcount_dims[1];
  nz = kt.kcount_dims[2] + 1;
  const int d0 = kt.kcount_dims[0], d1 = kt.kcount_dims[1], d2 = kt.kcount_dims[2];
  // planar vectors: origin + every non-z flat entry, sorted by |k_p|^2 so that a row tile is a ring of similar
  // radius and shares one kz cut-off (the listed k's fill a sphere: km_ewald.cpp:120-126)
  struct Pv { int ikx, iky, sgn, flat; double k2; };
  std::vector<Pv> pv;
  pv.push_back({0, 0, 1, -1, 0.0});
  for (int f = 0; f < kt.kcount_flat; ++f) {
    if (f >= d0 + d1 && f < d0 + d1 + d2) continue;  // z axis
    const int ax = std::abs(kt.kxvecs[f]), ay = std::abs(kt.kyvecs[f]);
    pv.push_back({ax, ay, kt.kyvecs[f] < 0 ? -1 : 1, f,
                  ax * ax * kt.unitk[0] * kt.unitk[0] + ay * ay * kt.unitk[1] * kt.unitk[1]});
  }
  std::stable_sort(pv.begin() + 1, pv.end(), [](const Pv &a, const Pv &b) { return a.k2 < b.k2; });
  // The vectors (kx, +ky) and (kx, -ky) share their x and y phase rows (a_+- = XrYr -+ XiYi, b_+- = XiYr +- XrYi).  Order: a head of
  // whole row tiles with the singles (the origin first, the axis vectors) filled up with the pairs of smallest |k_p|; then the other
  // pairs, each as (+, -) on an even index, by |k_p| -- the row tiles [paired_lo, paired_hi) hold 32 whole pairs each, zn_gemm forms both
  // members from one fetch of the two rows (conp_zn.hip); a last partial tile takes what is left.  Row tile after row tile the kz
  // range still shrinks (the head holds the origin: the full range), which the sk_gemm schedule counts on.
  {
    std::vector<Pv> pairs, singles;
    for (size_t i = 0; i < pv.size();) {
      if (i + 1 < pv.size() && pv[i].sgn > 0 && pv[i + 1].sgn < 0 && pv[i].ikx == pv[i + 1].ikx && pv[i].iky == pv[i + 1].iky &&
          pv[i].flat >= 0) {
        pairs.push_back(pv[i]); pairs.push_back(pv[i + 1]); i += 2;
      } else singles.push_back(pv[i++]);
    }
    std::vector<Pv> tail;
    if (singles.size() % 2) { tail.push_back(singles.back()); singles.pop_back(); }      // (an odd one out: the largest |k_p|, to the end)
    size_t fill = (PT - singles.size() % PT) % PT;                                         // vectors missing to whole tiles: even
    fill = std::min(fill, pairs.size());
    std::vector<Pv> out(singles);
    out.insert(out.end(), pairs.begin(), pairs.begin() + fill);
    paired_lo = paired_hi = (int)(out.size() / PT);
    if (out.size() % PT == 0) paired_hi += (int)((pairs.size() - fill) / PT);
    out.insert(out.end(), pairs.begin() + fill, pairs.end());
    out.insert(out.end(), tail.begin(), tail.end());
    pv.swap(out);
  }
  np = (int)pv.size();
  p_ikx.resize(np); p_iky.resize(np); p_sgn.resize(np);
  flat2p.assign(kt.kcount_flat, -1);
  for (int p = 0; p < np; ++p) {
    p_ikx[p] = pv[p].ikx; p_iky[p] = pv[p].iky; p_sgn[p] = pv[p].sgn;
    if (pv[p].flat >= 0) flat2p[pv[p].flat] = p;
  }
  // reference k index -> (p, m, sign)
  k_p.assign(kt.kcount, 0); k_m.assign(kt.kcount, 0); k_sign.assign(kt.kcount, 1);
  for (int f = 0; f < kt.kcount_flat; ++f) {
    if (flat2p[f] < 0) { k_p[f] = 0; k_m[f] = kt.kzvecs[f]; k_sign[f] = 1; }        // (the origin is the first vector)
    else { k_p[f] = flat2p[f]; k_m[f] = 0; k_sign[f] = 1; }
  }
  for (int e = 0; e < kt.kcount_expand; ++e) {
    const int k0 = kt.kcount_flat + 2 * e;
    const int p = flat2p[kt.kxy_list[e]];
    const int m = kt.kz_list[e] - (d0 + d1) + 1;
    k_p[k0] = p; k_m[k0] = m; k_sign[k0] = 1;
    k_p[k0 + 1] = p; k_m[k0 + 1] = m; k_sign[k0 + 1] = -1;
  }
  nblk = (nz + 15) / 16;
  n_col_tiles = (nblk + CT_BLK - 1) / CT_BLK;
  kzt = ((nz + n_col_tiles - 1) / n_col_tiles + 7) / 8 * 8;      // <= 160 because n_col_tiles = ceil(nz / 160)
  n_row_tiles = (np + PT - 1) / PT;
  R_pad = n_row_tiles * 2 * PT;
  C_pad = n_col_tiles * CT_COLS;
  // weights w(p,m) = sum over listed signs of 2 ug
  w.assign((size_t)np * nz, 0.0);
  for (int k = 0; k < kt.kcount; ++k) w[(size_t)k_p[k] * nz + k_m[k]] += 2.0 * kt.ug[k];
  wfull.assign((size_t)R_pad * C_pad, 0.0);
  nba_rc.assign((size_t)n_col_tiles * n_row_tiles, 0);
  nfa_fc.assign((size_t)n_col_tiles * n_row_tiles * 4, 0);
  for (int p = 0; p < np; ++p)
    for (int m = 0; m < nz; ++m) {
      const double ww = w[(size_t)p * nz + m];
      if (ww == 0.0) continue;
      wfull[(size_t)row_a(p) * C_pad + col_c(m)] = ww;
      wfull[(size_t)row_a(p) * C_pad + col_s(m)] = ww;
      wfull[(size_t)row_b(p) * C_pad + col_c(m)] = ww;
      wfull[(size_t)row_b(p) * C_pad + col_s(m)] = ww;
      const int ct = m / kzt, ml = m - ct * kzt;
      int &nb = nba_rc[(size_t)ct * n_row_tiles + p / PT], &nf = nfa_fc[(size_t)ct * n_row_tiles * 4 + p / 16];
      nb = std::max(nb, (ml >> 4) + 1);
      nf = std::max(nf, (ml >> 3) + 1);
    }
  sf_row_a.assign(kt.kcount, 0); sf_col_c.assign(kt.kcount, 0);
  for (int k = 0; k < kt.kcount; ++k) { sf_row_a[k] = row_a(k_p[k]); sf_col_c[k] = col_c(k_m[k]); }
}

// ------------------------------------------------------------------------------------------------
// PppmPlan
// ------------------------------------------------------------------------------------------------
void PppmPlan::rho1d(double dx, double *w) const {
  for (int k = 0; k < order; ++k) {
    double r = 0.0;
    for (int l = order - 1; l >= 0; --l) r = rho_coeff[(size_t)l * order + k] + r * dx;
    w[k] = r;
  }
}

void PppmPlan::build(int nx_, int ny_, int nz_, int order_, double g, double slabf, const double *lo, const double *prd) {
  nx = nx_; ny = ny_; nz = nz_; order = order_;
  if (order < 1 || order > MAXORDER) throw std::invalid_argument("pppm order out of range");
  nlower = -(order - 1) / 2; nupper = order / 2;
  if (order % 2) { shift = OFFSET + 0.5; shiftone = 0.0; } else { shift = OFFSET; shiftone = 0.5; }
  const double zprd_slab = prd[2] * slabf;
  for (int c = 0; c < 3; ++c) boxlo[c] = lo[c];
  volume = prd[0] * prd[1] * zprd_slab;
  delinv[0] = nx / prd[0]; delinv[1] = ny / prd[1]; delinv[2] = nz / zprd_slab;
  delvolinv = delinv[0] * delinv[1] * delinv[2];
  nfft = nx * ny * nz;
  // assignment-function polynomials (Hockney & Eastwood recursion as in PPPM::compute_rho_coeff)
  {
    const int W = 2 * order + 1;
    std::vector<double> a((size_t)order * W, 0.0);
    auto A = [&](int l, int k) -> double & { return a[(size_t)l * W + (k + order)]; };
    A(0, 0) = 1.0;
    for (int j = 1; j < order; ++j)
      for (int k = -j; k <= j; k += 2) {
        double sacc = 0.0;
        for (int l = 0; l < j; ++l) {
          A(l + 1, k) = (A(l, k + 1) - A(l, k - 1)) / (l + 1);
          sacc += std::pow(0.5, (double)l + 1) * (A(l, k - 1) + std::pow(-1.0, (double)l) * A(l, k + 1)) / (l + 1);
        }
        A(0, k) = sacc;
      }
    rho_coeff.assign((size_t)order * order, 0.0);
    int m = 0;
    for (int k = -(order - 1); k < order; k += 2, ++m)
      for (int l = 0; l < order; ++l) rho_coeff[(size_t)l * order + m] = A(l, k);
  }
  // denominator polynomial (PPPM::compute_gf_denom)
  double gf_b[MAXORDER];
  {
    for (int l = 1; l < order; ++l) gf_b[l] = 0.0;
    gf_b[0] = 1.0;
    for (int m = 1; m < order; ++m) {
      int l;
      for (l = m; l > 0; --l) gf_b[l] = 4.0 * (gf_b[l] * (l - m) * (l - m - 0.5) - gf_b[l - 1] * (l - m - 1) * (l - m - 1));
      gf_b[0] = 4.0 * (gf_b[0] * (l - m) * (l - m - 0.5));
    }
    long long ifact = 1;
    for (int k = 1; k < 2 * order; ++k) ifact *= k;
    const double gaminv = 1.0 / ifact;
    for (int l = 0; l < order; ++l) gf_b[l] *= gaminv;
  }
  auto gf_denom = [&](double x, double y, double z) {
    double sx = 0, sy = 0, sz = 0;
    for (int l = order - 1; l >= 0; --l) { sx = gf_b[l] + sx * x; sy = gf_b[l] + sy * y; sz = gf_b[l] + sz * z; }
    const double sprod = sx * sy * sz;
    return sprod * sprod;
  };
  auto powsinxx = [](double x, int n) { return x == 0.0 ? 1.0 : std::pow(std::sin(x) / x, n); };
  auto sq = [](double v) { return v * v; };
  // optimal influence function, ik differentiation (PPPM::compute_gf_ik)
  const double xprd = prd[0], yprd = prd[1];
  const double ukx = 2.0 * MY_PI / xprd, uky = 2.0 * MY_PI / yprd, ukz = 2.0 * MY_PI / zprd_slab;
  const double hoc = std::pow(-std::log(1.0e-7), 0.25);
  const int nbx = (int)((g * xprd / (MY_PI * nx)) * hoc), nby = (int)((g * yprd / (MY_PI * ny)) * hoc),
            nbz = (int)((g * zprd_slab / (MY_PI * nz)) * hoc);
  const int twoorder = 2 * order;
  greensfn.assign(nfft, 0.0);
  size_t n = 0;
  for (int m = 0; m < nz; ++m) {
    const int mper = m - nz * (2 * m / nz);
    const double snz = sq(std::sin(0.5 * ukz * mper * zprd_slab / nz));
    for (int l = 0; l < ny; ++l) {
      const int lper = l - ny * (2 * l / ny);
      const double sny = sq(std::sin(0.5 * uky * lper * yprd / ny));
      for (int k = 0; k < nx; ++k, ++n) {
        const int kper = k - nx * (2 * k / nx);
        const double snx = sq(std::sin(0.5 * ukx * kper * xprd / nx));
        const double sqk = sq(ukx * kper) + sq(uky * lper) + sq(ukz * mper);
        if (sqk == 0.0) continue;
        const double numerator = 12.5663706 / sqk;
        const double denominator = gf_denom(snx, sny, snz);
        double sum1 = 0.0;
        for (int ax = -nbx; ax <= nbx; ++ax) {
          const double qx = ukx * (kper + nx * ax), sx = std::exp(-0.25 * sq(qx / g)), wx = powsinxx(0.5 * qx * xprd / nx, twoorder);
          for (int ay = -nby; ay <= nby; ++ay) {
            const double qy = uky * (lper + ny * ay), sy = std::exp(-0.25 * sq(qy / g)), wy = powsinxx(0.5 * qy * yprd / ny, twoorder);
            for (int az = -nbz; az <= nbz; ++az) {
              const double qz = ukz * (mper + nz * az), sz = std::exp(-0.25 * sq(qz / g)), wz = powsinxx(0.5 * qz * zprd_slab / nz, twoorder);
              const double dot1 = ukx * kper * qx + uky * lper * qy + ukz * mper * qz;
              const double dot2 = qx * qx + qy * qy + qz * qz;
              sum1 += (dot1 / dot2) * sx * sy * sz * wx * wy * wz;
            }
          }
        }
        greensfn[n] = numerator * sum1 / denominator;
      }
    }
  }
  const int dims[3] = {nx, ny, nz};
  for (int c = 0; c < 3; ++c) {
    twid[c].resize(2 * (size_t)dims[c]);
    for (int t = 0; t < dims[c]; ++t) {
      twid[c][2 * t] = std::cos(2.0 * MY_PI * t / dims[c]);
      twid[c][2 * t + 1] = std::sin(2.0 * MY_PI * t / dims[c]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// electrode phase tables
// ------------------------------------------------------------------------------------------------
// The seeds of km_ewald.cpp:440-442: (cos, sin)(unitk_c * x_ic) for the three axes, [6][ne] = cx, sx, cy, sy, cz, sz.  Everything
// else of sincos_a_ele (the recurrences, the (kx, +-ky) products) runs on the device from these (conp_tables.hip).
// This function lives in THIS translation unit on purpose: it is compiled by g++ like the oracle and like a LAMMPS build of the
// reference, which turn the cos / sin pair of one argument into one sincos() call.  The same two lines compiled by hipcc's host
// clang call cos() and sin() separately, and glibc's sin() / cos() differ from its sincos() in the last bit for about 7 arguments
// in 10 000 (measured; on il_onelayer 186 of the 111 488 table entries moved) -- the tables would no longer be bit-identical to
// the reference's.
void electrode_seeds(const KTables &kt, int ne, const double *xele, std::vector<double> &seeds) {
  seeds.assign((size_t)6 * std::max(ne, 1), 0.0);
  for (int ic = 0; ic < 3; ++ic)
    for (int i = 0; i < ne; ++i) {
      const double xdotk = kt.unitk[ic] * xele[3 * (size_t)i + ic];
      seeds[(size_t)(2 * ic) * ne + i] = std::cos(xdotk);
      seeds[(size_t)(2 * ic + 1) * ne + i] = std::sin(xdotk);
    }
}

// ------------------------------------------------------------------------------------------------
// EleIndex
// ------------------------------------------------------------------------------------------------
void EleIndex::linalg_init(int nlocal, const int *tag, RankOps *ops) {   // fix_conp.cpp:413-416
  int maxtag = 0;
  for (int i = 0; i < nlocal; ++i) maxtag = std::max(tag[i], maxtag);
  if (ops) ops->allreduce_max_int(&maxtag, 1);                            // MPI_Allreduce MAX :415
  maxtag_all = maxtag;
  tag2eleall.assign((size_t)maxtag_all + 1, 0);
  elenum = elenum_all = elytenum = 0;
  initialised = true;
}

void EleIndex::map_atoms(int nlocal, const int *tag) {
  int maxtag = maxtag_all;
  for (int i = 0; i < nlocal; ++i) maxtag = std::max(maxtag, tag[i]);
  tag2local.assign((size_t)maxtag + 1, -1);
  for (int i = 0; i < nlocal; ++i) tag2local[tag[i]] = i;
}

bool EleIndex::post_neighbor(int nlocal, const int *tag, const int *echeck, bool *elyte_grew, RankOps *ops) {
  if (!initialised) throw std::logic_error("post_neighbor before linalg_init");
  RankOps one;
  if (!ops) ops = &one;
  const int nprocs = ops->nranks();
  const int elytenum_old = elytenum, elenum_all_old = elenum_all;
  // (the reference walks the owned atoms three times, :480-484, :493-498, :528-533; here once -- the electrode atoms' local indices
  //  in ascending order serve the other two walks)
  ele_local.clear();
  for (int i = 0; i < nlocal; ++i) if (echeck[i]) ele_local.push_back(i);
  elenum = (int)ele_local.size();
  elytenum = nlocal - elenum;
  ele2tag.resize(elenum);
  ele2eleall.resize(elenum);
  for (int j = 0; j < elenum; ++j) ele2tag[j] = tag[ele_local[j]];
  elenum_list.assign(nprocs, 0); displs.assign(nprocs, 0);
  ops->allgather_int(elenum, elenum_list.data());                // MPI_Allgather :492
  elenum_all = 0;
  for (int r = 0; r < nprocs; ++r) { displs[r] = elenum_all; elenum_all += elenum_list[r]; }   // :499-506
  const bool grew = elenum_all > elenum_all_old;
  if (grew) {                                                    // :510-525
    eleall2tag.assign(elenum_all, 0);
    ops->allgatherv_int(ele2tag.data(), elenum, eleall2tag.data(), elenum_list.data(), displs.data());   // :523
    elecheck_eleall.assign(elenum_all, 0);
    eleall2ele.assign((size_t)elenum_all + 1, -1);
    elebuf2eleall.assign(elenum_all, 0);
    std::fill(tag2eleall.begin(), tag2eleall.end(), elenum_all); // sentinel :521
    for (int i = 0; i < elenum_all; ++i) tag2eleall[eleall2tag[i]] = i;
  }
  for (int i = 0; i < elenum_all; ++i) eleall2ele[i] = -1;       // :527
  eleall2ele[elenum_all] = -1;
  for (int j = 0; j < elenum; ++j) {
    ele2eleall[j] = tag2eleall[tag[ele_local[j]]];
    eleall2ele[ele2eleall[j]] = j;
  }
  ops->allgatherv_int(ele2eleall.data(), elenum, elebuf2eleall.data(), elenum_list.data(), displs.data());   // :535
  if (elyte_grew) *elyte_grew = elytenum > elytenum_old;
  map_atoms(nlocal, tag);
  return grew;
}

void EleIndex::renumber_from_tags(const std::vector<int> &file_tags, int nlocal, const int *tag, const int *echeck, RankOps *ops) {
  if ((int)file_tags.size() != elenum_all) throw std::invalid_argument("matrix file tag row does not match the electrode count");
  eleall2tag = file_tags;
  for (int i = 0; i < elenum_all; ++i) {                 // :755-759
    eleall2ele[i] = -1;
    if (file_tags[i] < 0 || file_tags[i] > maxtag_all) throw std::invalid_argument("matrix file holds a tag outside 1..maxtag");
    tag2eleall[file_tags[i]] = i;
  }
  int j = 0;
  for (int i = 0; i < nlocal; ++i)                       // :763-770
    if (echeck[i]) {
      ele2tag[j] = tag[i];
      ele2eleall[j] = tag2eleall[tag[i]];
      eleall2ele[ele2eleall[j]] = j;
      ++j;
    }
  RankOps one;
  if (!ops) ops = &one;
  ops->allgatherv_int(ele2eleall.data(), elenum, elebuf2eleall.data(), elenum_list.data(), displs.data());
}

// ------------------------------------------------------------------------------------------------
// PairRows
// ------------------------------------------------------------------------------------------------
namespace {
struct RawPair { int row, ele, oth, col; };
void finish_rows(std::vector<RawPair> &raw, int ne, bool with_col, PairRows &out) {
  out.row_ptr.assign((size_t)ne + 1, 0);
  for (const auto &r : raw) ++out.row_ptr[r.row + 1];
  for (int i = 0; i < ne; ++i) out.row_ptr[i + 1] += out.row_ptr[i];
  std::vector<int> fill(out.row_ptr.begin(), out.row_ptr.end() - 1);
  out.ele_atom.resize(raw.size());
  out.oth_atom.resize(raw.size());
  out.col.resize(with_col ? raw.size() : 0);
  for (const auto &r : raw) {   // stable: list order is kept inside each row
    const int pos = fill[r.row]++;
    out.ele_atom[pos] = r.ele;
    out.oth_atom[pos] = r.oth;
    if (with_col) out.col[pos] = r.col;
  }
}
}  // namespace

// Counting sort of the listed pairs by electrode row, called at every re-neighbour with ~3e5 pairs at the decks' size.
// The row of a pair and its (electrode atom, partner) orientation follow fix_conp.cpp:1326-1350; list order is kept inside a row.
// The two passes (count, fill) cost ~3.5 ns per pair each on one core (random read-modify-writes), 1.9 ms at il_onelayer --
// several updates' worth -- so the owners are cut into contiguous ranges, one host thread each: thread t counts its range,
// the per-row offsets are stacked in thread order (= list order), every thread fills its own slots.  Same output as one thread.
static int host_threads() {
  static const int n = [] {
    if (const char *e = getenv("CONP_HOST_THREADS")) return std::max(1, atoi(e));
    const unsigned hw = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(8u, hw / 2));
  }();
  return n;
}

void build_b_rows(const ListView &l, int nlocal, const int *tag, const int *echeck, const EleIndex &idx, bool newton,
                  PairRows &out) {
  const int ne = idx.elenum_all;
  int nall = 0;
  size_t nlisted = 0;
  for (int ii = 0; ii < l.inum; ++ii) {
    const int i = l.ilist[ii];
    nall = std::max(nall, i + 1);
    const int *jlist = l.neigh + l.first[i];
    const int jnum = l.numneigh[i];
    nlisted += (size_t)jnum;
    for (int jj = 0; jj < jnum; ++jj) nall = std::max(nall, (jlist[jj] & NEIGHMASK) + 1);
  }
  static thread_local std::vector<int> arow_store, cnt_store;      // capacity survives between re-neighbours
  arow_store.resize(nall);
  int *const arow = arow_store.data();                             // arow[j] = eleall row of atom j, -1 for non-electrode atoms
  for (int i = 0; i < nall; ++i) arow[i] = echeck[i] ? idx.tag2eleall[tag[i]] : -1;
  // owner ranges of about equal pair counts
  int T = (nlisted < 20000) ? 1 : host_threads();
  std::vector<int> start(T + 1, l.inum);
  start[0] = 0;
  {
    size_t acc = 0;
    int t = 1;
    for (int ii = 0; ii < l.inum && t < T; ++ii) {
      acc += (size_t)l.numneigh[l.ilist[ii]];
      if (acc * T >= nlisted * (size_t)t) start[t++] = ii + 1;
    }
  }
  cnt_store.assign((size_t)T * (ne + 1), 0);                       // cnt[t][r] = pairs of row r found by thread t
  int *const cnt = cnt_store.data();
  const int nt = newton ? 1 : 0;
  auto qualifies_row = [&](int i, int ri, int j) {                 // row of the pair (owner i, neighbour j) or -1
    const int rj = arow[j];
    if (ri >= 0) return rj < 0 ? ri : -1;                          // :1339-1342
    return (rj >= 0 && (nt | (j < nlocal))) ? rj : -1;             // :1343-1350
  };
  auto count_range = [&](int t) {
    int *c = cnt + (size_t)t * (ne + 1);
    for (int ii = start[t]; ii < start[t + 1]; ++ii) {
      const int i = l.ilist[ii];
      const int ri = arow[i];
      const int *jlist = l.neigh + l.first[i];
      const int jnum = l.numneigh[i];
      for (int jj = 0; jj < jnum; ++jj) {
        const int r = qualifies_row(i, ri, jlist[jj] & NEIGHMASK);
        if (r >= 0) ++c[r];
      }
    }
  };
  auto run = [&](auto &&fn) {
    if (T == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(fn, t);
    fn(0);
    for (auto &x : th) x.join();
  };
  run(count_range);
  out.row_ptr.assign((size_t)ne + 1, 0);
  out.col.clear();
  for (int r = 0; r < ne; ++r) {                                   // stack the threads' counts of a row in thread order
    int pos = out.row_ptr[r];
    for (int t = 0; t < T; ++t) { const int c = cnt[(size_t)t * (ne + 1) + r]; cnt[(size_t)t * (ne + 1) + r] = pos; pos += c; }
    out.row_ptr[r + 1] = pos;
  }
  const size_t np = (size_t)out.row_ptr[ne];
  out.ele_atom.resize(np);
  out.oth_atom.resize(np);
  int *const ele = out.ele_atom.data(), *const oth = out.oth_atom.data();
  auto fill_range = [&](int t) {
    int *fill = cnt + (size_t)t * (ne + 1);                        // now: next free slot of this thread in each row
    for (int ii = start[t]; ii < start[t + 1]; ++ii) {
      const int i = l.ilist[ii];
      const int ri = arow[i];
      const int *jlist = l.neigh + l.first[i];
      const int jnum = l.numneigh[i];
      for (int jj = 0; jj < jnum; ++jj) {
        const int j = jlist[jj] & NEIGHMASK;
        const int r = qualifies_row(i, ri, j);
        if (r >= 0) {
          const int pos = fill[r]++;
          if (ri >= 0) { ele[pos] = i; oth[pos] = j; }
          else { ele[pos] = j; oth[pos] = i; }
        }
      }
    }
  };
  run(fill_range);
}

void build_pf_pairs(const ListView &l, const int *echeck, std::vector<int> &pi, std::vector<int> &pj) {
  size_t cap = 0;
  for (int ii = 0; ii < l.inum; ++ii) cap += (size_t)l.numneigh[l.ilist[ii]];
  pi.resize(cap); pj.resize(cap);
  size_t n = 0;
  for (int ii = 0; ii < l.inum; ++ii) {
    const int i = l.ilist[ii];
    const bool ei = echeck[i] != 0;
    const int *jlist = l.neigh + l.first[i];
    const int jnum = l.numneigh[i];
    for (int jj = 0; jj < jnum; ++jj) {
      const int j = jlist[jj] & NEIGHMASK;
      pi[n] = i; pj[n] = j;
      n += (size_t)(ei ^ (echeck[j] != 0));            // exactly one electrode member (fix_conp.cpp:1411)
    }
  }
  pi.resize(n); pj.resize(n);
}

void build_a_rows(const ListView &l, int nlocal, const int *tag, const int *echeck, const EleIndex &idx, bool newton,
                  PairRows &out) {
  std::vector<RawPair> raw;
  for (int ii = 0; ii < l.inum; ++ii) {
    const int i = l.ilist[ii];
    if (!echeck[i]) continue;
    const int *jlist = l.neigh + l.first[i];
    const int jnum = l.numneigh[i];
    const int elealli = idx.tag2eleall[tag[i]];
    for (int jj = 0; jj < jnum; ++jj) {
      const int j = jlist[jj] & NEIGHMASK;
      if (!echeck[j]) continue;                                   // :1255 ecib && ecjb
      const int eleallj = idx.tag2eleall[tag[j]];
      if (j < nlocal || !(!newton && eleallj > elealli))          // :1268
        raw.push_back({elealli, i, j, eleallj});
    }
  }
  finish_rows(raw, idx.elenum_all, true, out);
}

}  // namespace conp
