/* ---------------------------------------------------------------------------------------------
 * conp_oracle.c  --  TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * A plain-C, single-threaded CPU restatement of the constant-potential charge-solve hot path
 * of srtee/lammps-USER-CONP2 (fix_conp.cpp + km_ewald.cpp, low-memory Ewald provider, one MPI
 * rank).  It exists so that the HIP library can be checked against the reference's algorithm
 * on identical inputs, and so that bench.py can time a CPU baseline ("cpu_baseline.kind =
 * port").  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Every function cites the reference file:line whose arithmetic (operation order included) it
 * follows; citations are relative to /root/reference.  This file is C with its own data model
 * (flat arrays, CSR neighbour lists, an explicit "atoms view" instead of LAMMPS classes).  Where
 * bit-exactness with the reference depends on the ORDER of operations -- orc_inv_project,
 * orc_cg, the inner loop of orc_sincos_b -- the loops deliberately follow the cited lines
 * statement by statement, down to the reference's variable names, so that a reader can check
 * them side by side: those functions are a transliteration, not an independent derivation.
 * That is the point of an oracle; it is also why nothing outside tests/, smoke() and the
 * cpu_baseline leg may ever load this file.
 *
 * PARITY PIN (see DESIGN.md "Oracle"): the reference itself cannot be built in this image
 * (it needs the LAMMPS 27May2021 headers and library, which are absent, and writing stand-ins
 * for them is not allowed).  The oracle is therefore pinned by the only known-answer vector the
 * reference's tests hold for this path: tests/dilute/persist.log:112,143 (G vector
 * 0.77236341; step-0 electrode charge 0.044057154 / -0.044057154, |sum| < 1e-15, ffield etypes,
 * dV = 1 V) -- checked in tests/test_oracle_pin.py -- plus the deck-level invariants of
 * SURVEY.md section 4.
 * ------------------------------------------------------------------------------------------- */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef ORC_OPENMP
#include <omp.h>
#endif

/* LAMMPS math_const.h values (used at km_ewald.cpp:85-89, fix_conp.cpp:783) */
#define ORC_PI   3.14159265358979323846
#define ORC_PIS  1.77245385090551602729 /* sqrt(pi) */
#define ORC_4PI  12.56637061435917295384

/* erfc polynomial constants, fix_conp.cpp:53-60 */
#define ORC_EWALD_F 1.12837917
#define ORC_EWALD_P 0.3275911
#define ORC_A1 0.254829592
#define ORC_A2 (-0.284496736)
#define ORC_A3 1.421413741
#define ORC_A4 (-1.453152027)
#define ORC_A5 1.061405429
#define ORC_ERFC_MAX 5.8

#define ORC_NEIGHMASK 0x3FFFFFFF /* LAMMPS lmptype.h SBBITS=30 */

#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))

/* =============================================================================================
 * K-space tables  (km_ewald.cpp:63-132 conp_setup, :277-283 rms, :285-364 make_kvecs_ewald,
 * :366-381 make_ug_from_kvecs, :383-424 make_kxy_list_from_kvecs)
 * ===========================================================================================*/
typedef struct {
  double g_ewald, accuracy, slab_volfactor;
  int slabflag;
  double xprd, yprd, zprd, volume, gsqmx, ug_tot;
  double unitk[3];
  int kxmax, kymax, kzmax, kmax, kmax3d;
  int kcount, kcount_flat, kcount_expand;
  int kcount_dims[7];
  int *kxvecs, *kyvecs, *kzvecs, *kxy_list, *kz_list;
  double *ug;
} orc_kspace;

/* km_ewald.cpp:277-283 */
static double orc_rms(double g_ewald, int km, double prd, long long natoms, double q2) {
  double value = 2.0 * q2 * g_ewald / prd * sqrt(1.0 / (ORC_PI * km * natoms)) *
                 exp(-ORC_PI * ORC_PI * km * km / (g_ewald * g_ewald * prd * prd));
  return value;
}

/* enumerates the half-space k list in the reference order; if ks->kxvecs == NULL only counts.
 * km_ewald.cpp:285-361 */
static void orc_enumerate_kvecs(orc_kspace *ks) {
  int k, l, m, ic, kcount = 0;
  const int kmaxes[3] = {ks->kxmax, ks->kymax, ks->kzmax};
  double unitksq[3], sqk;
  int fill = ks->kxvecs != NULL;
  for (ic = 0; ic < 7; ++ic) ks->kcount_dims[ic] = 0;
  /* axes: (k,0,0) (0,l,0) (0,0,m)  :297-310 */
  for (ic = 0; ic < 3; ++ic) {
    unitksq[ic] = ks->unitk[ic] * ks->unitk[ic];
    for (m = 1; m <= kmaxes[ic]; ++m) {
      sqk = m * m * unitksq[ic];
      if (sqk <= ks->gsqmx) {
        if (fill) {
          if (ic == 0) ks->kxvecs[kcount] = m;
          else if (ic == 1) ks->kyvecs[kcount] = m;
          else ks->kzvecs[kcount] = m;
        }
        ++kcount;
        ++ks->kcount_dims[ic];
      }
    }
  }
  /* planes: (k,+-l,0) (0,k,+-l) (k,0,+-l)  :318-342 */
  for (ic = 3; ic < 6; ++ic) {
    int icA = (ic == 4) ? 1 : 0;
    int icB = (ic == 3) ? 1 : 2;
    for (k = 1; k <= kmaxes[icA]; ++k) {
      for (l = 1; l <= kmaxes[icB]; ++l) {
        sqk = k * k * unitksq[icA] + l * l * unitksq[icB];
        if (sqk <= ks->gsqmx) {
          if (fill) {
            int *va = (icA == 0) ? ks->kxvecs : ks->kyvecs;
            int *vb = (icB == 1) ? ks->kyvecs : ks->kzvecs;
            va[kcount] = k; vb[kcount] = l;
            va[kcount + 1] = k; vb[kcount + 1] = -l;
          }
          kcount += 2;
          ++ks->kcount_dims[ic];
        }
      }
    }
  }
  /* bulk: (k,+-l,+-m)  :346-359 */
  for (k = 1; k <= kmaxes[0]; ++k)
    for (l = 1; l <= kmaxes[1]; ++l)
      for (m = 1; m <= kmaxes[2]; ++m) {
        sqk = k * k * unitksq[0] + l * l * unitksq[1] + m * m * unitksq[2];
        if (sqk <= ks->gsqmx) {
          if (fill) {
            int s;
            for (s = 0; s < 4; ++s) {
              ks->kxvecs[kcount + s] = k;
              ks->kyvecs[kcount + s] = (s < 2) ? l : -l;
              ks->kzvecs[kcount + s] = (s % 2 == 0) ? m : -m;
            }
          }
          kcount += 4;
          ++ks->kcount_dims[6];
        }
      }
  ks->kcount = kcount;
  ks->kcount_flat = ks->kcount_dims[0] + ks->kcount_dims[1] + ks->kcount_dims[2] + 2 * ks->kcount_dims[3];
  ks->kcount_expand = ks->kcount_dims[4] + ks->kcount_dims[5] + 2 * ks->kcount_dims[6];
}

void orc_kspace_destroy(orc_kspace *ks) {
  if (!ks) return;
  free(ks->kxvecs); free(ks->kyvecs); free(ks->kzvecs);
  free(ks->kxy_list); free(ks->kz_list); free(ks->ug);
  free(ks);
}

/* qsqsum = sum of q^2 over ALL atoms at setup time (km_ewald.cpp:72-79); accuracy is the
 * ABSOLUTE force accuracy handed over by LAMMPS KSpace (relative accuracy * two_charge_force). */
orc_kspace *orc_kspace_create(double g_ewald, double accuracy, double slab_volfactor, int slabflag,
                              double xprd, double yprd, double zprd, double qsqsum,
                              long long natoms, double qqrd2e, double dielectric) {
  orc_kspace *ks = (orc_kspace *)calloc(1, sizeof(orc_kspace));
  double q2, zprd_slab, err, gsqxmx, gsqymx, gsqzmx;
  int k, kf, kxy, kloc, kx, ky;
  ks->g_ewald = g_ewald; ks->accuracy = accuracy; ks->slab_volfactor = slab_volfactor;
  ks->slabflag = slabflag; ks->xprd = xprd; ks->yprd = yprd; ks->zprd = zprd;
  q2 = qsqsum * qqrd2e / dielectric;                         /* :79 */
  zprd_slab = zprd * slab_volfactor;                         /* :84 */
  ks->volume = xprd * yprd * zprd_slab;                      /* :85 */
  ks->unitk[0] = 2.0 * ORC_PI / xprd;                        /* :87-89 */
  ks->unitk[1] = 2.0 * ORC_PI / yprd;
  ks->unitk[2] = 2.0 * ORC_PI / zprd_slab;
  ks->kxmax = ks->kymax = ks->kzmax = 1;                     /* :93-113 */
  err = orc_rms(g_ewald, ks->kxmax, xprd, natoms, q2);
  while (err > accuracy) { ks->kxmax++; err = orc_rms(g_ewald, ks->kxmax, xprd, natoms, q2); }
  err = orc_rms(g_ewald, ks->kymax, yprd, natoms, q2);
  while (err > accuracy) { ks->kymax++; err = orc_rms(g_ewald, ks->kymax, yprd, natoms, q2); }
  err = orc_rms(g_ewald, ks->kzmax, zprd_slab, natoms, q2);
  while (err > accuracy) { ks->kzmax++; err = orc_rms(g_ewald, ks->kzmax, zprd_slab, natoms, q2); }
  ks->kmax = ORC_MAX(ks->kxmax, ks->kymax);                  /* :115-117 */
  ks->kmax = ORC_MAX(ks->kmax, ks->kzmax);
  ks->kmax3d = 4 * ks->kmax * ks->kmax * ks->kmax + 6 * ks->kmax * ks->kmax + 3 * ks->kmax;
  gsqxmx = ks->unitk[0] * ks->unitk[0] * ks->kxmax * ks->kxmax;   /* :120-126 */
  gsqymx = ks->unitk[1] * ks->unitk[1] * ks->kymax * ks->kymax;
  gsqzmx = ks->unitk[2] * ks->unitk[2] * ks->kzmax * ks->kzmax;
  ks->gsqmx = ORC_MAX(gsqxmx, gsqymx);
  ks->gsqmx = ORC_MAX(ks->gsqmx, gsqzmx);
  ks->gsqmx *= 1.00001;
  /* the reference sizes every table by kmax3d; the oracle sizes by the actual count (two passes) */
  orc_enumerate_kvecs(ks);
  ks->kxvecs = (int *)calloc((size_t)ks->kcount + 4, sizeof(int));
  ks->kyvecs = (int *)calloc((size_t)ks->kcount + 4, sizeof(int));
  ks->kzvecs = (int *)calloc((size_t)ks->kcount + 4, sizeof(int));
  ks->ug = (double *)calloc((size_t)ks->kcount + 4, sizeof(double));
  orc_enumerate_kvecs(ks);
  { /* make_ug_from_kvecs :366-381 */
    double g_ewald_sq_inv = 1.0 / (g_ewald * g_ewald);
    double preu = 4.0 * ORC_PI / ks->volume, sqk;
    ks->ug_tot = 0;
    for (k = 0; k < ks->kcount; ++k) {
      sqk = ks->kxvecs[k] * ks->kxvecs[k] * ks->unitk[0] * ks->unitk[0];
      sqk += ks->kyvecs[k] * ks->kyvecs[k] * ks->unitk[1] * ks->unitk[1];
      sqk += ks->kzvecs[k] * ks->kzvecs[k] * ks->unitk[2] * ks->unitk[2];
      ks->ug[k] = preu * exp(-0.25 * sqk * g_ewald_sq_inv) / sqk;
      ks->ug_tot += 2 * ks->ug[k];
    }
  }
  /* make_kxy_list_from_kvecs :383-424 */
  ks->kxy_list = (int *)calloc((size_t)ks->kcount_expand + 2, sizeof(int));
  ks->kz_list = (int *)calloc((size_t)ks->kcount_expand + 2, sizeof(int));
  kf = ks->kcount_flat;
  for (k = 0; k < ks->kcount_dims[4]; ++k) {
    ks->kxy_list[k] = ks->kyvecs[kf] + ks->kcount_dims[0] - 1;
    ks->kz_list[k] = ks->kzvecs[kf] + ks->kcount_dims[0] + ks->kcount_dims[1] - 1;
    kf += 2;
  }
  for (k = ks->kcount_dims[4]; k < ks->kcount_dims[4] + ks->kcount_dims[5]; ++k) {
    ks->kxy_list[k] = ks->kxvecs[kf] - 1;
    ks->kz_list[k] = ks->kzvecs[kf] + ks->kcount_dims[0] + ks->kcount_dims[1] - 1;
    kf += 2;
  }
  kxy = ks->kcount_dims[0] + ks->kcount_dims[1] + ks->kcount_dims[2];
  kloc = ks->kcount_dims[4] + ks->kcount_dims[5];
  for (k = 0; k < ks->kcount_dims[6]; ++k) {
    kx = ks->kxvecs[kf]; ky = ks->kyvecs[kf];
    while (ks->kxvecs[kxy] != kx || ks->kyvecs[kxy] != ky) kxy += 2;
    ks->kxy_list[kloc] = kxy;
    ks->kxy_list[kloc + 1] = kxy + 1;
    ks->kz_list[kloc] = ks->kzvecs[kf] + ks->kcount_dims[0] + ks->kcount_dims[1] - 1;
    ks->kz_list[kloc + 1] = ks->kz_list[kloc];
    kf += 4;
    kloc += 2;
  }
  return ks;
}

/* thread count of the OpenMP build (bench.py cpu_baseline); no-op in the exact build */
int orc_set_threads(int n) {
#ifdef ORC_OPENMP
  omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

/* getters (ctypes-friendly) */
void orc_kspace_info(const orc_kspace *ks, int *iout /*[16]*/, double *dout /*[8]*/) {
  int i;
  iout[0] = ks->kcount; iout[1] = ks->kcount_flat; iout[2] = ks->kcount_expand;
  iout[3] = ks->kxmax; iout[4] = ks->kymax; iout[5] = ks->kzmax; iout[6] = ks->kmax; iout[7] = ks->kmax3d;
  for (i = 0; i < 7; ++i) iout[8 + i] = ks->kcount_dims[i];
  iout[15] = ks->slabflag;
  dout[0] = ks->unitk[0]; dout[1] = ks->unitk[1]; dout[2] = ks->unitk[2];
  dout[3] = ks->volume; dout[4] = ks->gsqmx; dout[5] = ks->ug_tot; dout[6] = ks->g_ewald; dout[7] = ks->accuracy;
}
void orc_kspace_tables(const orc_kspace *ks, int *kx, int *ky, int *kz, double *ug, int *kxy_list, int *kz_list) {
  memcpy(kx, ks->kxvecs, sizeof(int) * ks->kcount);
  memcpy(ky, ks->kyvecs, sizeof(int) * ks->kcount);
  memcpy(kz, ks->kzvecs, sizeof(int) * ks->kcount);
  memcpy(ug, ks->ug, sizeof(double) * ks->kcount);
  memcpy(kxy_list, ks->kxy_list, sizeof(int) * ks->kcount_expand);
  memcpy(kz_list, ks->kz_list, sizeof(int) * ks->kcount_expand);
}

/* =============================================================================================
 * Structure factors of the electrolyte  (km_ewald.cpp:668-780 sincos_b)
 * x: [n][3] AoS, q: [n], echeck: [n] (+1 group1, -1 group2, 0 electrolyte; fix_conp.cpp:599-605).
 * Only atoms 0..nlocal-1 with echeck==0 && q!=0 enter (:686).  Output sfacrl/sfacim[kcount].
 * Work tables cs/sn are [kcount_flat][jmax], k-major / atom-contiguous like the reference.
 * Returns jmax.
 * ===========================================================================================*/
int orc_sincos_b(const orc_kspace *ks, int nlocal, const double *x, const double *q, const int *echeck,
                 double *sfacrl, double *sfacim) {
  int i, j, m, ic, kf, jmax = 0;
  const int kflat = ks->kcount_flat;
  double *cs, *sn, *qj;
  const int *kcd = ks->kcount_dims;
  for (i = 0; i < nlocal; ++i) if (echeck[i] == 0 && q[i] != 0) ++jmax;
  memset(sfacrl, 0, sizeof(double) * ks->kcount);   /* km_ewald.cpp:160-161 (there: kmax3d) */
  memset(sfacim, 0, sizeof(double) * ks->kcount);
  if (jmax == 0) return 0;
  cs = (double *)malloc(sizeof(double) * (size_t)kflat * jmax);
  sn = (double *)malloc(sizeof(double) * (size_t)kflat * jmax);
  qj = (double *)malloc(sizeof(double) * jmax);
#define CS(k, jj) cs[(size_t)(k) * jmax + (jj)]
#define SN(k, jj) sn[(size_t)(k) * jmax + (jj)]
  j = 0;
  for (i = 0; i < nlocal; ++i) {                      /* :685-697 */
    if (echeck[i] == 0 && q[i] != 0) {
      qj[j] = q[i];
      kf = 0;
      for (ic = 0; ic < 3; ++ic) {
        double xdotk = ks->unitk[ic] * x[3 * i + ic];
        CS(kf, j) = cos(xdotk);
        SN(kf, j) = sin(xdotk);
        kf += kcd[ic];
      }
      ++j;
    }
  }
  kf = 0;
  for (ic = 0; ic < 3; ++ic) {                        /* :699-724 axis recurrences */
    double tr = 0, ti = 0;
    for (j = 0; j < jmax; ++j) { tr += qj[j] * CS(kf, j); ti += qj[j] * SN(kf, j); }
    sfacrl[kf] = tr; sfacim[kf] = ti;
    for (m = 1; m < kcd[ic]; ++m) {
      double *c1 = &CS(kf + m, 0), *s1 = &SN(kf + m, 0);
      const double *c0 = &CS(kf + m - 1, 0), *s0 = &SN(kf + m - 1, 0);
      const double *cb = &CS(kf, 0), *sb = &SN(kf, 0);
      tr = 0; ti = 0;
      for (j = 0; j < jmax; ++j) {
        c1[j] = c0[j] * cb[j] - s0[j] * sb[j];
        s1[j] = s0[j] * cb[j] + c0[j] * sb[j];
        tr += qj[j] * c1[j];
        ti += qj[j] * s1[j];
      }
      sfacrl[kf + m] = tr; sfacim[kf + m] = ti;
    }
    kf += kcd[ic];
  }
  for (m = 0; m < kcd[3]; ++m) {                      /* :728-754 (k,+-l,0) */
    int kx = ks->kxvecs[kf] - 1;
    int ky = ks->kyvecs[kf] + kcd[0] - 1;
    double tr0 = 0, ti0 = 0, tr1 = 0, ti1 = 0;
    double *c0 = &CS(kf, 0), *s0 = &SN(kf, 0), *c1 = &CS(kf + 1, 0), *s1 = &SN(kf + 1, 0);
    const double *cx = &CS(kx, 0), *sx = &SN(kx, 0), *cy = &CS(ky, 0), *sy = &SN(ky, 0);
    for (j = 0; j < jmax; ++j) {
      c0[j] = cx[j] * cy[j] - sx[j] * sy[j];
      s0[j] = cx[j] * sy[j] + sx[j] * cy[j];
      tr0 += qj[j] * c0[j];
      ti0 += qj[j] * s0[j];
      c1[j] = cx[j] * cy[j] + sx[j] * sy[j];
      s1[j] = -cx[j] * sy[j] + sx[j] * cy[j];
      tr1 += qj[j] * c1[j];
      ti1 += qj[j] * s1[j];
    }
    sfacrl[kf] = tr0; sfacim[kf] = ti0; sfacrl[kf + 1] = tr1; sfacim[kf + 1] = ti1;
    kf += 2;
  }
#ifdef ORC_OPENMP
#pragma omp parallel for schedule(static) private(j)
#endif
  for (m = 0; m < ks->kcount_expand; ++m) {           /* :761-779 (..,+-m) pairs */
    const int kf = ks->kcount_flat + 2 * m;
    const double *cxy = &CS(ks->kxy_list[m], 0), *sxy = &SN(ks->kxy_list[m], 0);
    const double *cz = &CS(ks->kz_list[m], 0), *sz = &SN(ks->kz_list[m], 0);
    double tr0 = 0, ti0 = 0, tr1 = 0, ti1 = 0;
    for (j = 0; j < jmax; ++j) {
      tr0 += qj[j] * (cxy[j] * cz[j] - sxy[j] * sz[j]);
      ti0 += qj[j] * (cxy[j] * sz[j] + sxy[j] * cz[j]);
      tr1 += qj[j] * (cxy[j] * cz[j] + sxy[j] * sz[j]);
      ti1 += qj[j] * (-cxy[j] * sz[j] + sxy[j] * cz[j]);
    }
    sfacrl[kf] = tr0; sfacim[kf] = ti0; sfacrl[kf + 1] = tr1; sfacim[kf + 1] = ti1;
  }
#undef CS
#undef SN
  free(cs); free(sn); free(qj);
  return jmax;
}

/* =============================================================================================
 * Electrode phase tables, low-memory form  (km_ewald.cpp:426-475 sincos_a_ele, lowmem branch;
 * :510-531 transposition into csk/snk[eleall][kcount_flat])
 * xele: [ne][3] coordinates in eleall order.  csk/snk: [ne][kflat] atom-major.
 * ===========================================================================================*/
void orc_ele_trig(const orc_kspace *ks, int ne, const double *xele, double *csk, double *snk) {
  const int kflat = ks->kcount_flat;
  const int *kcd = ks->kcount_dims;
  int i, m, ic, kf;
  for (i = 0; i < ne; ++i) {
    double *c = csk + (size_t)i * kflat, *s = snk + (size_t)i * kflat;
    kf = 0;
    for (ic = 0; ic < 3; ++ic) {
      double xdotk = ks->unitk[ic] * xele[3 * i + ic];
      c[kf] = cos(xdotk);
      s[kf] = sin(xdotk);
      for (m = 1; m < kcd[ic]; ++m) {                 /* :452-458 */
        c[kf + m] = c[kf + m - 1] * c[kf] - s[kf + m - 1] * s[kf];
        s[kf + m] = s[kf + m - 1] * c[kf] + c[kf + m - 1] * s[kf];
      }
      kf += kcd[ic];
    }
    for (m = 0; m < kcd[3]; ++m) {                    /* :464-477 */
      int kx = ks->kxvecs[kf] - 1;
      int ky = ks->kyvecs[kf] + kcd[0] - 1;
      c[kf] = c[kx] * c[ky] - s[kx] * s[ky];
      s[kf] = c[kx] * s[ky] + s[kx] * c[ky];
      c[kf + 1] = c[kx] * c[ky] + s[kx] * s[ky];
      s[kf + 1] = -c[kx] * s[ky] + s[kx] * c[ky];
      kf += 2;
    }
  }
}

/* km_ewald.cpp:533-558 kz_expand: regenerate the 2*kexp expanded phases of one electrode atom */
static void orc_kz_expand(const orc_kspace *ks, const double *c, const double *s, double *ce, double *se) {
  int k;
  for (k = 0; k < ks->kcount_expand; ++k) {
    double cxy = c[ks->kxy_list[k]], sxy = s[ks->kxy_list[k]];
    double cz = c[ks->kz_list[k]], sz = s[ks->kz_list[k]];
    ce[2 * k] = cxy * cz - sxy * sz;
    se[2 * k] = sxy * cz + cxy * sz;
    ce[2 * k + 1] = cxy * cz + sxy * sz;
    se[2 * k + 1] = sxy * cz - cxy * sz;
  }
}

/* =============================================================================================
 * k-space b projection, lowmem  (km_ewald.cpp:789-825 bbb_from_sincos_b, else-branch)
 * bbb[ne] is overwritten, in eleall order.
 * ===========================================================================================*/
void orc_bbb_from_sincos_b(const orc_kspace *ks, int ne, const double *csk, const double *snk,
                           const double *sfacrl, const double *sfacim, double *bbb) {
  const int kflat = ks->kcount_flat, kexp = ks->kcount_expand;
  int i, k;
#ifdef ORC_OPENMP
#pragma omp parallel private(i, k)
#endif
  {
  double *ce = (double *)malloc(sizeof(double) * 2 * (kexp + 1));
  double *se = (double *)malloc(sizeof(double) * 2 * (kexp + 1));
#ifdef ORC_OPENMP
#pragma omp for schedule(static)
#endif
  for (i = 0; i < ne; ++i) {
    const double *c = csk + (size_t)i * kflat, *s = snk + (size_t)i * kflat;
    double bbbtmp = 0;
    orc_kz_expand(ks, c, s, ce, se);
    for (k = 0; k < kflat; ++k) bbbtmp -= 2 * ks->ug[k] * (c[k] * sfacrl[k] + s[k] * sfacim[k]);
    for (k = 0; k < 2 * kexp; ++k)
      bbbtmp -= 2 * ks->ug[kflat + k] * (ce[k] * sfacrl[kflat + k] + se[k] * sfacim[kflat + k]);
    bbb[i] = bbbtmp;
  }
  free(ce); free(se);
  }
}

/* km_ewald.cpp:827-847 slabcorr: bbb[i] -= z_i * sum_{j not electrode} 4 pi q_j z_j / V */
double orc_slabcorr(const orc_kspace *ks, int nlocal, const double *x, const double *q, const int *echeck,
                    int ne, const double *xele, double *bbb) {
  double slabcorr = 0.0;
  int i;
  for (i = 0; i < nlocal; ++i)
    if (echeck[i] == 0) slabcorr += 4 * q[i] * ORC_PI * x[3 * i + 2] / ks->volume;
  for (i = 0; i < ne; ++i) bbb[i] -= xele[3 * i + 2] * slabcorr;
  return slabcorr;
}

/* =============================================================================================
 * k-space part of A, lowmem, single rank  (km_ewald.cpp:584-666 aaa_from_sincos_a else-branch,
 * :560-582 ewald_dot_ij).  aaa: [ne][ne] row = eleall index (one rank: ele == eleall).
 * Fills ONE orientation of each unordered pair by the parity rule (:625-640), the diagonal
 * (:631-634) and the slab term for j <= i (:647-665).  Caller zeroes aaa first (fix_conp.cpp:792).
 * ===========================================================================================*/
void orc_aaa_from_sincos_a(const orc_kspace *ks, int ne, const double *csk, const double *snk,
                           const double *xele, double *aaa) {
  const int kflat = ks->kcount_flat, kexp = ks->kcount_expand;
  const double CON_2overPIS = 2.0 / ORC_PIS;
  double *cie = (double *)malloc(sizeof(double) * 2 * (kexp + 1));
  double *sie = (double *)malloc(sizeof(double) * 2 * (kexp + 1));
  double *cje = (double *)malloc(sizeof(double) * 2 * (kexp + 1));
  double *sje = (double *)malloc(sizeof(double) * 2 * (kexp + 1));
  int i, j, k, pass;
  for (i = 0; i < ne; ++i) {
    const double *ci = csk + (size_t)i * kflat, *si = snk + (size_t)i * kflat;
    orc_kz_expand(ks, ci, si, cie, sie);
    for (pass = 0; pass < 2; ++pass) {
      int j0 = pass == 0 ? i % 2 : i + 1, j1 = pass == 0 ? i : ne;
      for (j = j0; j < j1; j += 2) {
        const double *cj = csk + (size_t)j * kflat, *sj = snk + (size_t)j * kflat;
        double aaatmp = 0;
        orc_kz_expand(ks, cj, sj, cje, sje);
        for (k = 0; k < kflat; ++k) aaatmp += 2 * ks->ug[k] * (ci[k] * cj[k] + si[k] * sj[k]);
        for (k = 0; k < 2 * kexp; ++k) aaatmp += 2 * ks->ug[kflat + k] * (cie[k] * cje[k] + sie[k] * sje[k]);
        aaa[(size_t)i * ne + j] = aaatmp;
      }
    }
    aaa[(size_t)i * ne + i] = ks->ug_tot - CON_2overPIS * ks->g_ewald;
  }
  if (ks->slabflag == 1) {
    double CON_4PIoverV = ORC_4PI / ks->volume;
    for (i = 0; i < ne; ++i)
      for (j = 0; j <= i; ++j) aaa[(size_t)i * ne + j] += CON_4PIoverV * xele[3 * i + 2] * xele[3 * j + 2];
  }
  free(cie); free(sie); free(cje); free(sje);
}

/* =============================================================================================
 * Real-space kernels  (fix_conp.cpp:1446-1454 erfcr_sqrt, :1467-1475 eta_potential_A / eta_potential)
 * ===========================================================================================*/
double orc_erfcr_sqrt(double a2_r2) {
  if (a2_r2 < ORC_ERFC_MAX * ORC_ERFC_MAX) {
    double a_r = sqrt(a2_r2);
    double expm2 = exp(-a2_r2);
    double t = 1.0 / (1.0 + ORC_EWALD_P * a_r);
    return t * (ORC_A1 + t * (ORC_A2 + t * (ORC_A3 + t * (ORC_A4 + t * ORC_A5)))) * expm2 / a_r;
  }
  return 0.;
}
static double orc_eta_potential_A(double eta, double rsq) {
  double etarij2 = eta * eta * rsq / 2;
  return -orc_erfcr_sqrt(etarij2) * eta / sqrt(2);
}
static double orc_eta_potential(double eta, double rsq) {
  double etarij2 = eta * eta * rsq;
  return -orc_erfcr_sqrt(etarij2) * eta;
}

/* LAMMPS-style half neighbour list in CSR form: for ii<inum, i=ilist[ii], neighbours are
 * neigh[first[i] .. first[i]+numneigh[i]-1] (entries may carry special-bond bits). */
typedef struct {
  int inum;
  const int *ilist, *numneigh, *first, *neigh;
} orc_list;

typedef struct {
  int nlocal, nghost, ntypes;
  const double *x;     /* [nall][3] */
  double *q;           /* [nall]  (electrode entries are overwritten by update_charge) */
  const int *type, *tag, *echeck; /* [nall] */
} orc_atoms;

/* =============================================================================================
 * FixConp restatement, one MPI rank
 * ===========================================================================================*/
typedef struct {
  /* command / force-field parameters */
  double eta, tolerance, evscale, g_ewald, cut_coul;
  int ff_flag;      /* 0 NORMAL(slab), 1 FFIELD, 2 NOSLAB   fix_conp.cpp:68 */
  int zneutr, nullneutral, minimizer /*0 CG, 1 INV*/, maxiter, newton, qinit, one_electrode;
  int ntypes;
  double *cutsq;    /* [(ntypes+1)^2] */
  double boxlo_z, zprd;
  orc_kspace *ks;
  /* EHGO pair mode (fix_conp.cpp:1482-1598): per-type eta_i, u0_i (already * evscale), kappa -> eta_ij, fo_ij */
  int ehgo;
  double kappa, *eta_i, *u0_i, *eta_ij, *fo_ij;
  /* bookkeeping (fix_conp.cpp:468-539) */
  int elenum, elenum_all, elytenum, maxtag_all;
  int *ele2tag, *ele2eleall, *tag2eleall, *eleall2tag, *eleall2ele, *elecheck_eleall, *elebuf2eleall;
  int *tag2local, tag2local_n;
  /* linear algebra state */
  double *aaa_all, *bbb_all, *eleallq, *elesetq, *eleinitq, *bbb;
  double *csk, *snk, *xele_all;
  double totsetq, scalar_output, totinve_e, slabcorr_last;
  int runstage, cg_iters;
  orc_atoms at;
  orc_list alist, blist;
} orc_fix;

orc_fix *orc_fix_create(double eta, int ff_flag, int zneutr, int nullneutral, int minimizer, int maxiter,
                        double tolerance, int newton, int qinit, int one_electrode, double evscale,
                        int ntypes, const double *cutsq, double cut_coul, double boxlo_z, double zprd,
                        orc_kspace *ks) {
  orc_fix *f = (orc_fix *)calloc(1, sizeof(orc_fix));
  size_t nc = (size_t)(ntypes + 1) * (ntypes + 1);
  f->eta = eta; f->ff_flag = ff_flag; f->zneutr = zneutr; f->nullneutral = nullneutral;
  f->minimizer = minimizer; f->maxiter = maxiter; f->tolerance = tolerance; f->newton = newton;
  f->qinit = qinit; f->one_electrode = one_electrode; f->evscale = evscale; f->ntypes = ntypes;
  f->cutsq = (double *)malloc(sizeof(double) * nc);
  memcpy(f->cutsq, cutsq, sizeof(double) * nc);
  f->cut_coul = cut_coul; f->boxlo_z = boxlo_z; f->zprd = zprd; f->ks = ks;
  f->g_ewald = ks->g_ewald;
  return f;
}

void orc_fix_destroy(orc_fix *f) {
  if (!f) return;
  free(f->eta_i); free(f->u0_i); free(f->eta_ij); free(f->fo_ij);
  free(f->cutsq); free(f->ele2tag); free(f->ele2eleall); free(f->tag2eleall); free(f->eleall2tag);
  free(f->eleall2ele); free(f->elecheck_eleall); free(f->elebuf2eleall); free(f->tag2local);
  free(f->aaa_all); free(f->bbb_all); free(f->eleallq); free(f->elesetq); free(f->eleinitq); free(f->bbb);
  free(f->csk); free(f->snk); free(f->xele_all);
  free(f);
}

/* fix_modify ... ehgo kappa / coeff (fix_conp.cpp:1482-1515) already parsed: per-type arrays [ntypes+1], u0 in eV/e^2
 * (or auto = sqrt(2) eta / sqrt(pi) / evscale), then ehgo_setup_tables (:1517-1559).  Returns 0 if no coefficient is set
 * (the reference then falls back to ETA with a warning). */
int orc_fix_set_ehgo(orc_fix *f, double kappa, const double *eta_i, const double *u0_ev) {
  const int nt1 = f->ntypes + 1;
  const double CON_s2overPIS = sqrt(2.0) / ORC_PIS, sq8 = sqrt(8.0);
  double *f_i = (double *)calloc(nt1, sizeof(double));
  int i, j, setflag = 0;
  f->kappa = kappa;
  f->eta_i = (double *)calloc(nt1, sizeof(double)); f->u0_i = (double *)calloc(nt1, sizeof(double));
  f->eta_ij = (double *)calloc((size_t)nt1 * nt1, sizeof(double)); f->fo_ij = (double *)calloc((size_t)nt1 * nt1, sizeof(double));
  for (i = 1; i < nt1; ++i) { f->eta_i[i] = eta_i[i]; f->u0_i[i] = u0_ev[i] * f->evscale; if (f->eta_i[i] || f->u0_i[i]) setflag = 1; }
  if (setflag) {
    for (i = 1; i < nt1; ++i) f_i[i] = f->u0_i[i] - CON_s2overPIS * f->eta_i[i];
    for (i = 1; i < nt1; ++i)
      for (j = 1; j <= i; ++j) {
        if (f->eta_i[i] && f->eta_i[j]) {
          double etasq = f->eta_i[i] * f->eta_i[i] + f->eta_i[j] * f->eta_i[j];
          double etaprod = f->eta_i[i] * f->eta_i[j];
          double eij = etaprod / sqrt(etasq);
          double o_ij = sq8 * eij * eij * eij / (etaprod * sqrt(etaprod));
          double f_ij = 0.5 * kappa * (f_i[i] + f_i[j]);
          f->eta_ij[i * nt1 + j] = eij;
          f->fo_ij[i * nt1 + j] = f_ij * o_ij;
        } else f->eta_ij[i * nt1 + j] = f->eta_i[i] + f->eta_i[j];
        if (i != j) { f->eta_ij[j * nt1 + i] = f->eta_ij[i * nt1 + j]; f->fo_ij[j * nt1 + i] = f->fo_ij[i * nt1 + j]; }
      }
    f->ehgo = 1;
  }
  free(f_i);
  return setflag;
}

/* fix_conp.cpp:1561-1573 */
static double orc_ehgo_potential(const orc_fix *f, double rsq, int itype, int jtype) {
  double etaij = f->eta_ij[itype * (f->ntypes + 1) + jtype];
  double foij = f->fo_ij[itype * (f->ntypes + 1) + jtype];
  double etarij2 = etaij * etaij * rsq;
  return foij * exp(-0.5 * etarij2) - orc_erfcr_sqrt(etarij2) * etaij;
}

void orc_fix_set_atoms(orc_fix *f, int nlocal, int nghost, const double *x, double *q, const int *type,
                       const int *tag, const int *echeck) {
  int i, maxtag = 0;
  f->at.nlocal = nlocal; f->at.nghost = nghost; f->at.x = x; f->at.q = q; f->at.type = type;
  f->at.tag = tag; f->at.echeck = echeck; f->at.ntypes = f->ntypes;
  for (i = 0; i < nlocal; ++i) maxtag = ORC_MAX(maxtag, tag[i]);
  if (maxtag + 1 > f->tag2local_n) {
    f->tag2local = (int *)realloc(f->tag2local, sizeof(int) * (maxtag + 1));
    f->tag2local_n = maxtag + 1;
  }
  for (i = 0; i < f->tag2local_n; ++i) f->tag2local[i] = -1;
  for (i = 0; i < nlocal; ++i) f->tag2local[tag[i]] = i;  /* atom->map(tag) for owned atoms */
}

void orc_fix_set_lists(orc_fix *f, int a_inum, const int *a_ilist, const int *a_numneigh, const int *a_first,
                       const int *a_neigh, int b_inum, const int *b_ilist, const int *b_numneigh,
                       const int *b_first, const int *b_neigh) {
  f->alist.inum = a_inum; f->alist.ilist = a_ilist; f->alist.numneigh = a_numneigh;
  f->alist.first = a_first; f->alist.neigh = a_neigh;
  f->blist.inum = b_inum; f->blist.ilist = b_ilist; f->blist.numneigh = b_numneigh;
  f->blist.first = b_first; f->blist.neigh = b_neigh;
}

/* fix_conp.cpp:468-539 post_neighbor (nprocs == 1: displs = {0}, elebuf2eleall = ele2eleall) plus
 * the tag2eleall sizing of linalg_init (:413-416). */
void orc_fix_post_neighbor(orc_fix *f) {
  const orc_atoms *at = &f->at;
  int i, j, elenum = 0, elenum_all_old = f->elenum_all;
  if (f->tag2eleall == NULL) { /* linalg_init, runstage 0 */
    int maxtag = 0;
    for (i = 0; i < at->nlocal; ++i) maxtag = ORC_MAX(at->tag[i], maxtag);
    f->maxtag_all = maxtag;
    f->tag2eleall = (int *)malloc(sizeof(int) * (maxtag + 1));
  }
  for (i = 0; i < at->nlocal; ++i) if (at->echeck[i]) ++elenum;
  f->elytenum = at->nlocal - elenum;
  if (elenum > f->elenum || f->ele2tag == NULL) {
    f->ele2tag = (int *)realloc(f->ele2tag, sizeof(int) * (elenum + 1));
    f->ele2eleall = (int *)realloc(f->ele2eleall, sizeof(int) * (elenum + 1));
    f->bbb = (double *)realloc(f->bbb, sizeof(double) * (elenum + 1));
  }
  f->elenum = elenum;
  j = 0;
  for (i = 0; i < at->nlocal; ++i) if (at->echeck[i]) f->ele2tag[j++] = at->tag[i];
  f->elenum_all = elenum; /* single rank: displssum */
  if (f->elenum_all > elenum_all_old) {
    size_t n = (size_t)f->elenum_all;
    f->eleall2tag = (int *)realloc(f->eleall2tag, sizeof(int) * n);
    f->elecheck_eleall = (int *)realloc(f->elecheck_eleall, sizeof(int) * n);
    f->eleall2ele = (int *)realloc(f->eleall2ele, sizeof(int) * (n + 1));
    f->aaa_all = (double *)realloc(f->aaa_all, sizeof(double) * n * n);
    f->bbb_all = (double *)realloc(f->bbb_all, sizeof(double) * n);
    f->eleallq = (double *)realloc(f->eleallq, sizeof(double) * n);
    f->elebuf2eleall = (int *)realloc(f->elebuf2eleall, sizeof(int) * n);
    f->elesetq = (double *)realloc(f->elesetq, sizeof(double) * n);
    f->eleinitq = (double *)realloc(f->eleinitq, sizeof(double) * n);
    for (i = 0; i < f->elenum_all; i++) f->elecheck_eleall[i] = 0;
    for (i = 0; i < f->maxtag_all + 1; i++) f->tag2eleall[i] = f->elenum_all;
    f->eleall2ele[f->elenum_all] = -1;
    for (i = 0; i < f->elenum_all; ++i) f->eleall2tag[i] = f->ele2tag[i]; /* Allgatherv of one rank */
    for (i = 0; i < f->elenum_all; ++i) f->tag2eleall[f->eleall2tag[i]] = i;
  }
  j = 0;
  for (i = 0; i < f->elenum_all; ++i) f->eleall2ele[i] = -1;
  for (i = 0; i < at->nlocal; ++i) {
    if (at->echeck[i]) {
      f->ele2eleall[j] = f->tag2eleall[at->tag[i]];
      f->eleall2ele[f->ele2eleall[j]] = j;
      ++j;
    }
  }
  for (i = 0; i < f->elenum; ++i) f->elebuf2eleall[i] = f->ele2eleall[i];
}

/* fix_conp.cpp:641-648 b_comm with one rank: brecv[elebuf2eleall[i]] = bsend[i] */
static void orc_b_comm(const orc_fix *f, const double *bsend, double *brecv) {
  int iall;
  for (iall = 0; iall < f->elenum_all; ++iall) brecv[f->elebuf2eleall[iall]] = bsend[iall];
}

/* gathers electrode coordinates in eleall order (what sincos_a_ele + b_comm achieve, km_ewald.cpp:437) */
static void orc_gather_xele(orc_fix *f) {
  int i, c;
  f->xele_all = (double *)realloc(f->xele_all, sizeof(double) * 3 * (size_t)f->elenum_all);
  for (i = 0; i < f->elenum; ++i) {
    int iloc = f->tag2local[f->ele2tag[i]];
    for (c = 0; c < 3; ++c) f->xele_all[3 * f->ele2eleall[i] + c] = f->at.x[3 * iloc + c];
  }
}

/* fix_conp.cpp:1209-1279 alist_coul_cal: ele-ele real-space pairs into m[elei*Ne + eleallj] */
static void orc_alist_coul_cal(orc_fix *f, double *m) {
  const orc_atoms *at = &f->at;
  const orc_list *L = &f->alist;
  const int nt1 = f->ntypes + 1;
  int ii, jj;
  double cut_coulsq = f->cut_coul * f->cut_coul;
  double cut_erfc = ORC_ERFC_MAX * ORC_ERFC_MAX / (f->g_ewald * f->g_ewald);
  if (cut_coulsq > cut_erfc) cut_coulsq = cut_erfc;
  for (ii = 0; ii < L->inum; ii++) {
    int i = L->ilist[ii];
    int itype = at->type[i];
    int ecib = !!at->echeck[i];
    double xtmp = at->x[3 * i], ytmp = at->x[3 * i + 1], ztmp = at->x[3 * i + 2];
    const int *jlist = L->neigh + L->first[i];
    int jnum = L->numneigh[i];
    for (jj = 0; jj < jnum; jj++) {
      int j = jlist[jj] & ORC_NEIGHMASK;
      int ecjb = !!at->echeck[j];
      if (ecib && ecjb) {
        double delx = xtmp - at->x[3 * j], dely = ytmp - at->x[3 * j + 1], delz = ztmp - at->x[3 * j + 2];
        double rsq = delx * delx + dely * dely + delz * delz;
        int jtype = at->type[j];
        if (rsq < f->cutsq[itype * nt1 + jtype]) {
          if (rsq < cut_coulsq) {
            double dudq = orc_erfcr_sqrt(f->g_ewald * f->g_ewald * rsq) * f->g_ewald;
            int elealli, eleallj, elei;
            dudq += f->ehgo ? orc_ehgo_potential(f, rsq, itype, jtype) : orc_eta_potential_A(f->eta, rsq);
            elealli = f->tag2eleall[at->tag[i]];
            eleallj = f->tag2eleall[at->tag[j]];
            elei = f->eleall2ele[elealli];
            if (j < at->nlocal || !(!f->newton && eleallj > elealli))
              m[(size_t)elei * f->elenum_all + eleallj] += dudq;
          }
        }
      }
    }
  }
}

/* fix_conp.cpp:1281-1365 blist_coul_cal: ele-elyte real-space pairs into m[elei] (local ele order) */
static void orc_blist_coul_cal(orc_fix *f, double *m) {
  const orc_atoms *at = &f->at;
  const orc_list *L = &f->blist;
  const int nt1 = f->ntypes + 1;
  int ii, jj, elei;
  double cut_coulsq = f->cut_coul * f->cut_coul;
  double cut_erfc = ORC_ERFC_MAX * ORC_ERFC_MAX / (f->g_ewald * f->g_ewald);
  double *newtonbuf = NULL;
  if (cut_coulsq > cut_erfc) cut_coulsq = cut_erfc;
  if (f->newton) newtonbuf = (double *)calloc(f->elenum_all, sizeof(double));
  for (ii = 0; ii < L->inum; ii++) {
    int i = L->ilist[ii];
    int ecib = at->echeck[i] != 0;
    int itype = at->type[i];
    double xtmp = at->x[3 * i], ytmp = at->x[3 * i + 1], ztmp = at->x[3 * i + 2];
    const int *jlist = L->neigh + L->first[i];
    int jnum = L->numneigh[i];
    for (jj = 0; jj < jnum; jj++) {
      int j = jlist[jj] & ORC_NEIGHMASK;
      int ecjb = at->echeck[j] != 0;
      if ((ecib ^ ecjb) && (f->newton || ecib || j < at->nlocal)) {
        double delx = xtmp - at->x[3 * j], dely = ytmp - at->x[3 * j + 1], delz = ztmp - at->x[3 * j + 2];
        double rsq = delx * delx + dely * dely + delz * delz;
        int jtype = at->type[j];
        if (rsq < f->cutsq[itype * nt1 + jtype]) {
          if (rsq < cut_coulsq) {
            double dudq = orc_erfcr_sqrt(f->g_ewald * f->g_ewald * rsq) * f->g_ewald;
            dudq += f->ehgo ? orc_ehgo_potential(f, rsq, itype, jtype) : orc_eta_potential(f->eta, rsq);
            if (ecib) {
              elei = f->eleall2ele[f->tag2eleall[at->tag[i]]];
              m[elei] -= at->q[j] * dudq;
            } else if (j < at->nlocal) {
              int elej = f->eleall2ele[f->tag2eleall[at->tag[j]]];
              m[elej] -= at->q[i] * dudq;
            } else if (f->newton) {
              newtonbuf[f->tag2eleall[at->tag[j]]] -= at->q[i] * dudq;
            }
          }
        }
      }
    }
  }
  if (f->newton) {
    for (elei = 0; elei < f->elenum; ++elei) m[elei] += newtonbuf[f->tag2eleall[f->ele2tag[elei]]];
    free(newtonbuf);
  }
}

/* fix_conp.cpp:777-861 a_cal (one rank) */
void orc_fix_a_cal(orc_fix *f) {
  const int ne = f->elenum_all;
  const double CON_s2overPIS = sqrt(2.0) / ORC_PIS;
  double *aaa = (double *)calloc((size_t)f->elenum * ne, sizeof(double));
  double *aaa_perm;
  int i, j;
  /* kspmod->a_cal: a_read (electrode tables) + aaa_from_sincos_a.  The k-space routine works in
   * eleall numbering; with one rank row "i" of aaa is local electrode i -> permute rows. */
  orc_gather_xele(f);
  f->csk = (double *)realloc(f->csk, sizeof(double) * (size_t)ne * f->ks->kcount_flat);
  f->snk = (double *)realloc(f->snk, sizeof(double) * (size_t)ne * f->ks->kcount_flat);
  orc_ele_trig(f->ks, ne, f->xele_all, f->csk, f->snk);
  aaa_perm = (double *)calloc((size_t)ne * ne, sizeof(double));
  orc_aaa_from_sincos_a(f->ks, ne, f->csk, f->snk, f->xele_all, aaa_perm);
  for (i = 0; i < f->elenum; ++i)
    memcpy(aaa + (size_t)i * ne, aaa_perm + (size_t)f->ele2eleall[i] * ne, sizeof(double) * ne);
  free(aaa_perm);
  for (i = 0; i < f->elenum; ++i) /* :796-810 */
    aaa[(size_t)i * ne + f->ele2eleall[i]] += f->ehgo ? f->u0_i[f->at.type[f->tag2local[f->ele2tag[i]]]] : CON_s2overPIS * f->eta;
  orc_alist_coul_cal(f, aaa);
  /* Allgatherv of rows (:816-822): one rank, rows are in local ele order -> rank-major == local */
  memcpy(f->aaa_all, aaa, sizeof(double) * (size_t)f->elenum * ne);
  free(aaa);
  for (i = 1; i < ne; ++i)                              /* :826-831 symmetrise */
    for (j = 0; j < i; ++j) {
      f->aaa_all[(size_t)i * ne + j] += f->aaa_all[(size_t)j * ne + i];
      f->aaa_all[(size_t)j * ne + i] = f->aaa_all[(size_t)i * ne + j];
    }
  f->runstage = 1;
}

/* fix_conp.cpp:609-637 b_setq_cal */
void orc_fix_b_setq_cal(orc_fix *f) {
  const orc_atoms *at = &f->at;
  double zlo = f->boxlo_z, zprd = f->zprd;
  double zprd_half = 0.5 * zprd; /* domain->zprd_half */
  double zhalf = zprd_half + zlo;
  int iloc;
  for (iloc = 0; iloc < f->elenum_all; ++iloc) f->elecheck_eleall[iloc] = 0;
  for (iloc = 0; iloc < f->elenum; ++iloc) {
    int iall = f->ele2eleall[iloc];
    int i = f->tag2local[f->ele2tag[iloc]];
    int eci = at->echeck[i];
    if (f->ff_flag == 1) {
      if (eci == 1 && at->x[3 * i + 2] < zhalf) f->bbb[iloc] = -f->evscale * (at->x[3 * i + 2] / zprd + 1);
      else f->bbb[iloc] = -f->evscale * at->x[3 * i + 2] / zprd;
    } else f->bbb[iloc] = -0.5 * f->evscale * eci;
    f->elecheck_eleall[iall] = eci;
  }
  orc_b_comm(f, f->bbb, f->bbb_all);
  if (f->runstage == 1) f->runstage = 2;
}

/* Row-major LU inverse with partial pivoting (stands in for dgetrf_/dgetri_, fix_conp.cpp:947-949;
 * LAPACK's blocked operation order is vendor-specific and not reproduced). Returns 0 on success. */
int orc_lu_inverse(int n, double *a) {
  int *piv = (int *)malloc(sizeof(int) * n);
  double *inv = (double *)malloc(sizeof(double) * (size_t)n * n);
  double *col = (double *)malloc(sizeof(double) * n);
  int i, j, k, info = 0;
  for (k = 0; k < n; ++k) {
    int p = k;
    double big = fabs(a[(size_t)k * n + k]);
    for (i = k + 1; i < n; ++i) if (fabs(a[(size_t)i * n + k]) > big) { big = fabs(a[(size_t)i * n + k]); p = i; }
    piv[k] = p;
    if (big == 0.0) { info = k + 1; break; }
    if (p != k) for (j = 0; j < n; ++j) { double t = a[(size_t)k * n + j]; a[(size_t)k * n + j] = a[(size_t)p * n + j]; a[(size_t)p * n + j] = t; }
    for (i = k + 1; i < n; ++i) {
      double l = a[(size_t)i * n + k] / a[(size_t)k * n + k];
      double *ri = a + (size_t)i * n;
      const double *rk = a + (size_t)k * n;
      ri[k] = l;
      for (j = k + 1; j < n; ++j) ri[j] -= l * rk[j];
    }
  }
  if (info == 0) {
    /* solve A X = I column by column with the permutation applied to the right-hand side */
    for (j = 0; j < n; ++j) {
      for (i = 0; i < n; ++i) col[i] = 0.0;
      col[j] = 1.0;
      for (k = 0; k < n; ++k) if (piv[k] != k) { double t = col[k]; col[k] = col[piv[k]]; col[piv[k]] = t; }
      for (i = 0; i < n; ++i) { double s = col[i]; const double *ri = a + (size_t)i * n; for (k = 0; k < i; ++k) s -= ri[k] * col[k]; col[i] = s; }
      for (i = n - 1; i >= 0; --i) { double s = col[i]; const double *ri = a + (size_t)i * n; for (k = i + 1; k < n; ++k) s -= ri[k] * col[k]; col[i] = s / ri[i]; }
      for (i = 0; i < n; ++i) inv[(size_t)i * n + j] = col[i];
    }
    memcpy(a, inv, sizeof(double) * (size_t)n * n);
  }
  free(piv); free(inv); free(col);
  return info;
}

/* fix_conp.cpp:982-1067 inv_project.  eleallz: z of each electrode atom in eleall order (zneutr).
 * Operates in place on aaa[n*n]; returns totinve of the first projection (the <e,e> log value / evscale). */
double orc_inv_project(int n, double *aaa, int nullneutral, int zneutr, const double *eleallz, double zhalf) {
  double *ainve = (double *)malloc(sizeof(double) * n);
  double ainvtmp, totinve = 0, totinve_first;
  size_t idx1d = 0;
  int i, j;
  for (i = 0; i < n; i++) {
    ainvtmp = 0;
    for (j = 0; j < n; j++) { ainvtmp += aaa[idx1d]; idx1d++; }
    totinve += ainvtmp;
    ainve[i] = ainvtmp;
  }
  totinve_first = totinve;
  if (nullneutral) {
    if (totinve * totinve > 1e-8) {
      idx1d = 0;
      for (i = 0; i < n; i++)
        for (j = 0; j < n; j++) { aaa[idx1d] -= ainve[i] * ainve[j] / totinve; idx1d++; }
    }
    if (zneutr) {
      idx1d = 0;
      totinve = 0;
      for (i = 0; i < n; i++) {
        ainvtmp = 0;
        for (j = 0; j < n; j++) { if (eleallz[j] > zhalf) ainvtmp += aaa[idx1d]; idx1d++; }
        ainve[i] = ainvtmp;
        if (eleallz[i] > zhalf) totinve += ainvtmp;
      }
      if (totinve * totinve > 1e-8) {
        idx1d = 0;
        for (i = 0; i < n; i++)
          for (j = 0; j < n; j++) { aaa[idx1d] -= ainve[i] * ainve[j] / totinve; idx1d++; }
      }
    }
  }
  free(ainve);
  return totinve_first;
}

static void orc_fix_inv_project(orc_fix *f) {
  int i;
  double *z = (double *)malloc(sizeof(double) * f->elenum_all);
  orc_gather_xele(f);
  for (i = 0; i < f->elenum_all; ++i) z[i] = f->xele_all[3 * i + 2];
  f->totinve_e = orc_inv_project(f->elenum_all, f->aaa_all, f->nullneutral, f->zneutr, z,
                                 0.5 * f->zprd + f->boxlo_z);
  free(z);
}

/* fix_conp.cpp:932-980 inv */
int orc_fix_inv(orc_fix *f) {
  int info = 0;
  if (f->runstage == 2) {
    info = orc_lu_inverse(f->elenum_all, f->aaa_all);
    if (info == 0 && !f->one_electrode) orc_fix_inv_project(f);
    f->runstage = 3;
  }
  return info;
}

/* fix_conp.cpp:864-930 cg: neutrality-constrained CG on the un-projected A, x0 = 0 */
int orc_cg(int n, const double *aaa, const double *bbb, double *q, int maxiter, double tolerance) {
  double *res = (double *)malloc(sizeof(double) * n), *p = (double *)malloc(sizeof(double) * n),
         *ap = (double *)malloc(sizeof(double) * n);
  double alpha, beta, ptap, lresnorm, netr, tmp, lgamma, gamma, avenetr;
  int iter, i, j, converged_at = 0;
  for (i = 0; i < n; i++) q[i] = 0.0;
  lresnorm = 0.0;
  netr = 0.0;
  for (i = 0; i < n; ++i) {
    res[i] = bbb[i];
    for (j = 0; j < n; ++j) { tmp = aaa[(size_t)i * n + j] * q[j]; res[i] -= tmp; }
    netr += res[i];
    lresnorm += res[i] * res[i];
  }
  avenetr = netr / n;
  for (i = 0; i < n; i++) p[i] = res[i] - avenetr;
  lresnorm -= netr * avenetr;
  lgamma = lresnorm;
  for (iter = 1; iter < maxiter; ++iter) {
    for (i = 0; i < n; ++i) {
      ap[i] = 0.0;
      for (j = 0; j < n; ++j) ap[i] += aaa[(size_t)i * n + j] * p[j];
    }
    ptap = 0.0;
    for (i = 0; i < n; ++i) ptap += p[i] * ap[i];
    alpha = lresnorm / ptap;
    gamma = lgamma;
    lgamma = 0.0;
    netr = 0.0;
    for (i = 0; i < n; ++i) {
      q[i] = q[i] + alpha * p[i];
      res[i] = res[i] - alpha * ap[i];
      lgamma += res[i] * res[i];
      netr += res[i];
    }
    avenetr = netr / n;
    lgamma -= netr * avenetr;
    beta = lgamma / gamma;
    lresnorm = 0.0;
    for (i = 0; i < n; i++) {
      p[i] = beta * p[i] + res[i] - avenetr;
      lresnorm += res[i] * p[i];
    }
    if (lresnorm / n < tolerance) { converged_at = iter; break; }
  }
  free(res); free(p); free(ap);
  return converged_at; /* 0 = hit maxiter */
}

/* fix_conp.cpp:698-718 equation_solve */
int orc_fix_equation_solve(orc_fix *f) {
  if (f->minimizer == 0) {
    f->cg_iters = orc_cg(f->elenum_all, f->aaa_all, f->bbb_all, f->eleallq, f->maxiter, f->tolerance);
    return 0;
  }
  return orc_fix_inv(f);
}

/* row dot product standing in for BLAS ddot_ (fix_conp.cpp:1093,1138): plain left-to-right sum */
static double orc_ddot(int n, const double *a, const double *b) {
  double s = 0;
  int i;
  for (i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* the row-dot GEMV of update_charge in isolation (fix_conp.cpp:1135-1139), for bench.py's cpu_baseline leg */
void orc_gemv_rows(int n, const double *aaa, const double *b, double *y) {
  int i;
#ifdef ORC_OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (i = 0; i < n; ++i) y[i] = orc_ddot(n, aaa + (size_t)i * n, b);
}

/* fix_conp.cpp:1071-1116 get_setq */
void orc_fix_get_setq(orc_fix *f) {
  int iall, iloc;
  if (f->minimizer == 0) {
    for (iall = 0; iall < f->elenum_all; ++iall) f->elesetq[iall] = f->eleallq[iall];
  } else {
    for (iloc = 0; iloc < f->elenum; ++iloc) {
      iall = f->ele2eleall[iloc];
      f->bbb[iloc] = orc_ddot(f->elenum_all, f->aaa_all + (size_t)iall * f->elenum_all, f->bbb_all);
    }
    orc_b_comm(f, f->bbb, f->elesetq);
  }
  f->totsetq = 0;
  for (iloc = 0; iloc < f->elenum; ++iloc) {
    iall = f->ele2eleall[iloc];
    if (f->elecheck_eleall[iall] == 1) f->totsetq += f->elesetq[iall];
  }
  if (f->qinit) {
    for (iloc = 0; iloc < f->elenum; ++iloc) f->bbb[iloc] = f->at.q[f->tag2local[f->ele2tag[iloc]]];
    orc_b_comm(f, f->bbb, f->eleinitq);
  }
  if (f->one_electrode) orc_fix_inv_project(f);
}

/* fix_conp.cpp:426-464 linalg_setup (runstage 0): a_cal, b_setq_cal, equation_solve, get_setq */
int orc_fix_linalg_setup(orc_fix *f) {
  int info;
  orc_fix_a_cal(f);
  orc_fix_b_setq_cal(f);
  info = orc_fix_equation_solve(f);
  orc_fix_get_setq(f);
  return info;
}

/* fix_conp.cpp:677-695 update_bk / km_ewald.cpp:153-167 b_cal */
void orc_fix_b_cal(orc_fix *f, int coulyes) {
  const orc_kspace *ks = f->ks;
  const orc_atoms *at = &f->at;
  double *sr = (double *)malloc(sizeof(double) * (ks->kcount + 1));
  double *si = (double *)malloc(sizeof(double) * (ks->kcount + 1));
  double *ball = (double *)malloc(sizeof(double) * (f->elenum_all + 1));
  int i;
  orc_sincos_b(ks, at->nlocal, at->x, at->q, at->echeck, sr, si);
  orc_bbb_from_sincos_b(ks, f->elenum_all, f->csk, f->snk, sr, si, ball); /* eleall order */
  if (ks->slabflag) f->slabcorr_last = orc_slabcorr(ks, at->nlocal, at->x, at->q, at->echeck, f->elenum_all, f->xele_all, ball);
  for (i = 0; i < f->elenum; ++i) f->bbb[i] = ball[f->ele2eleall[i]];     /* local ele order */
  if (coulyes) orc_blist_coul_cal(f, f->bbb);
  orc_b_comm(f, f->bbb, f->bbb_all);
  free(sr); free(si); free(ball);
}

/* fix_conp.cpp:1120-1161 update_charge */
void orc_fix_update_charge(orc_fix *f, double potdiff) {
  const orc_atoms *at = &f->at;
  int i, iall, iloc;
  const int nall = at->nlocal + at->nghost;
  double netcharge_left = 0;
  if (f->minimizer == 1) {
    for (iloc = 0; iloc < f->elenum; ++iloc) {
      iall = f->ele2eleall[iloc];
      f->bbb[iloc] = orc_ddot(f->elenum_all, f->aaa_all + (size_t)iall * f->elenum_all, f->bbb_all);
    }
    orc_b_comm(f, f->bbb, f->eleallq);
  }
  for (iall = 0; iall < f->elenum_all; ++iall)
    if (f->elecheck_eleall[iall] == 1) netcharge_left += f->eleallq[iall];
  for (i = 0; i < nall; ++i) {
    if (!at->echeck[i]) continue;
    iall = f->tag2eleall[at->tag[i]];
    at->q[i] = f->eleallq[iall] + potdiff * f->elesetq[iall];
    if (f->qinit) at->q[i] += f->eleinitq[iall];
  }
  f->scalar_output = potdiff * f->totsetq + netcharge_left;
}

/* fix_conq.cpp:41-90 FixConq::update_charge: potential difference that puts total charge -Q / +Q on group1 / group2.
 * rightcharge = the DV argument of `fix conq`.  Returns the potential difference used (= the fix scalar). */
double orc_fix_update_charge_conq(orc_fix *f, double rightcharge) {
  const orc_atoms *at = &f->at;
  int i, iall, iloc;
  const int nall = at->nlocal + at->nghost;
  double netcharge_right = 0, potdiff_conq;
  if (f->minimizer == 1) {
    for (iloc = 0; iloc < f->elenum; ++iloc) {
      iall = f->ele2eleall[iloc];
      f->bbb[iloc] = orc_ddot(f->elenum_all, f->aaa_all + (size_t)iall * f->elenum_all, f->bbb_all);
    }
    orc_b_comm(f, f->bbb, f->eleallq);
  }
  for (iall = 0; iall < f->elenum_all; ++iall)
    if (f->elecheck_eleall[iall] == 1) netcharge_right -= f->eleallq[iall];
  f->scalar_output = -(rightcharge - netcharge_right) / f->totsetq;
  if (f->one_electrode) f->scalar_output += 2 * rightcharge / f->totsetq;
  potdiff_conq = f->scalar_output;
  for (i = 0; i < nall; ++i) {
    if (!at->echeck[i]) continue;
    iall = f->tag2eleall[at->tag[i]];
    at->q[i] = f->eleallq[iall] + potdiff_conq * f->elesetq[iall];
    if (f->qinit) at->q[i] += f->eleinitq[iall];
  }
  return potdiff_conq;
}

/* fix_cond.cpp:46-126 FixCond: cond_setup (setzvec = preset vector / evscale, taken right after b_setq_cal), cond_setup2
 * (vmult) and update_charge.  xprd/yprd: box lengths (Axy).  Returns the potential difference (= fix scalar). */
double orc_fix_update_charge_cond(orc_fix *f, double rightcharge, double xprd, double yprd, const double *setzvec) {
  const orc_atoms *at = &f->at;
  const double lz = f->zprd, Axy = xprd * yprd;
  double zOAz = 0., vmult, dipole = 0., potdiff;
  int i, iall, iloc;
  for (i = 0; i < f->elenum_all; ++i) zOAz += f->elesetq[i] * setzvec[i];            /* cond_setup2 :58-68 */
  vmult = 4 * ORC_PI * zOAz * lz / (f->evscale * Axy);
  vmult /= 1 + vmult;
  vmult /= zOAz;
  if (f->minimizer == 1) {
    for (iloc = 0; iloc < f->elenum; ++iloc) {
      iall = f->ele2eleall[iloc];
      f->bbb[iloc] = orc_ddot(f->elenum_all, f->aaa_all + (size_t)iall * f->elenum_all, f->bbb_all);
    }
    orc_b_comm(f, f->bbb, f->eleallq);
  }
  for (i = 0; i < at->nlocal; ++i) if (!at->echeck[i]) dipole -= at->q[i] * at->x[3 * i + 2];   /* :101-106 */
  potdiff = rightcharge - dipole / lz;
  for (iall = 0; iall < f->elenum_all; ++iall) potdiff -= setzvec[iall] * f->eleallq[iall];
  potdiff *= vmult;
  f->scalar_output = potdiff;
  for (iall = 0; iall < f->elenum_all; ++iall) {                                        /* :118-124: owned atoms via atom->map */
    i = f->tag2local[f->eleall2tag[iall]];
    if (i != -1) {
      at->q[i] = f->eleallq[iall] + potdiff * f->elesetq[iall];
      if (f->qinit) at->q[i] += f->eleinitq[iall];
    }
  }
  return potdiff;
}

/* fix_conp.cpp:1456-1465 ferfcr_sqrt, :1477-1480 eta_force */
static double orc_ferfcr_sqrt(double a2_r2) {
  if (a2_r2 < ORC_ERFC_MAX * ORC_ERFC_MAX) {
    double a_r = sqrt(a2_r2);
    double expm2 = exp(-a2_r2);
    double t = 1.0 / (1.0 + ORC_EWALD_P * a_r);
    double erfcr = t * (ORC_A1 + t * (ORC_A2 + t * (ORC_A3 + t * (ORC_A4 + t * ORC_A5)))) * expm2 / a_r;
    return erfcr + ORC_EWALD_F * expm2;
  }
  return 0.;
}

/* fix_conp.cpp:1163-1201 force_cal (ETA pair mode) + :1368-1444 blist_coul_cal_post_force.
 * fadd[nall][3] is ACCUMULATED into (like atom->f); out[0] = kspace energy increment (:1167-1181),
 * out[1] = eng_coul increment, out[2..7] = virial increment in LAMMPS order xx,yy,zz,xy,xz,yz as Pair::ev_tally adds them
 * for eflag_global/vflag_global (pair.cpp ev_tally, newton_pair rule).  The reference applies del*forcecoul (not
 * del*forcecoul/rsq) to the forces and gates on eta^2 r^2 < 5.8 (not 5.8^2) -- restated as written (:1418-1431). */
void orc_fix_post_force(orc_fix *f, double qqrd2e, double *fadd, double *out) {
  const orc_atoms *at = &f->at;
  const orc_list *L = &f->blist;
  const int nt1 = f->ntypes + 1;
  int i, ii, jj, k;
  double eleqsqsum = 0.0;
  for (k = 0; k < 8; ++k) out[k] = 0.0;
  if (!f->ehgo) {
    for (i = 0; i < at->nlocal; i++)
      if (at->echeck[i]) eleqsqsum += at->q[i] * at->q[i];
    out[0] = qqrd2e * 1.0 * f->eta * eleqsqsum / (sqrt(2) * ORC_PIS);
  } else { /* :1182-1199 */
    for (i = 0; i < at->nlocal; i++)
      if (at->echeck[i]) eleqsqsum += f->u0_i[at->type[i]] * at->q[i] * at->q[i];
    out[0] = qqrd2e * 1.0 * eleqsqsum;
  }
  for (ii = 0; ii < L->inum; ii++) {
    int i = L->ilist[ii];
    int eleilocal = !!at->echeck[i];
    double qtmp = at->q[i];
    double xtmp = at->x[3 * i], ytmp = at->x[3 * i + 1], ztmp = at->x[3 * i + 2];
    int itype = at->type[i];
    const int *jlist = L->neigh + L->first[i];
    int jnum = L->numneigh[i];
    for (jj = 0; jj < jnum; jj++) {
      int j = jlist[jj] & ORC_NEIGHMASK;
      int elejlocal = !!at->echeck[j];
      if (eleilocal ^ elejlocal) {
        double delx = xtmp - at->x[3 * j], dely = ytmp - at->x[3 * j + 1], delz = ztmp - at->x[3 * j + 2];
        double rsq = delx * delx + dely * dely + delz * delz;
        int jtype = at->type[j];
        if (rsq < f->cutsq[itype * nt1 + jtype]) {
          double etarij2 = f->eta * f->eta * rsq;
          if (etarij2 < ORC_ERFC_MAX) {
            double prefactor = qqrd2e * qtmp * at->q[j];
            double forcecoul;
            if (!f->ehgo) forcecoul = prefactor * (-orc_ferfcr_sqrt(etarij2) * f->eta);
            else { /* ehgo_force :1568-1573 */
              double etaij = f->eta_ij[itype * nt1 + jtype], foij = f->fo_ij[itype * nt1 + jtype];
              double e2 = etaij * etaij * rsq;
              forcecoul = prefactor * (e2 * foij * exp(-0.5 * e2) - orc_ferfcr_sqrt(e2) * etaij);
            }
            double fpair = forcecoul / rsq;
            double ecoul, v[6], w = 0.0;
            if (!eleilocal) {
              fadd[3 * i] += delx * forcecoul; fadd[3 * i + 1] += dely * forcecoul; fadd[3 * i + 2] += delz * forcecoul;
            } else if (f->newton || j < at->nlocal) {
              fadd[3 * j] -= delx * forcecoul; fadd[3 * j + 1] -= dely * forcecoul; fadd[3 * j + 2] -= delz * forcecoul;
            }
            ecoul = prefactor * (f->ehgo ? orc_ehgo_potential(f, rsq, itype, jtype) : orc_eta_potential(f->eta, rsq));
            /* Pair::ev_tally global accumulators */
            if (f->newton) w = 1.0;
            else { if (i < at->nlocal) w += 0.5; if (j < at->nlocal) w += 0.5; }
            out[1] += w * ecoul;
            v[0] = delx * delx * fpair; v[1] = dely * dely * fpair; v[2] = delz * delz * fpair;
            v[3] = delx * dely * fpair; v[4] = delx * delz * fpair; v[5] = dely * delz * fpair;
            for (k = 0; k < 6; ++k) out[2 + k] += w * v[k];
          }
        }
      }
    }
  }
}

/* fix_conp.cpp:543-573 pre_force (the every-Nevery gate is the caller's) */
void orc_fix_pre_force(orc_fix *f, double potdiff) {
  orc_fix_b_cal(f, 1);
  orc_fix_equation_solve(f);
  orc_fix_update_charge(f, potdiff);
}

/* getters */
void orc_fix_sizes(const orc_fix *f, int *out /*[8]*/) {
  out[0] = f->elenum; out[1] = f->elenum_all; out[2] = f->elytenum; out[3] = f->maxtag_all;
  out[4] = f->runstage; out[5] = f->cg_iters; out[6] = 0; out[7] = 0;
}
void orc_fix_scalars(const orc_fix *f, double *out /*[4]*/) {
  out[0] = f->totsetq; out[1] = f->scalar_output; out[2] = f->totinve_e; out[3] = f->slabcorr_last;
}
void orc_fix_get_maps(const orc_fix *f, int *ele2tag, int *ele2eleall, int *eleall2tag, int *eleall2ele,
                      int *elecheck_eleall, int *elebuf2eleall, int *tag2eleall) {
  memcpy(ele2tag, f->ele2tag, sizeof(int) * f->elenum);
  memcpy(ele2eleall, f->ele2eleall, sizeof(int) * f->elenum);
  memcpy(eleall2tag, f->eleall2tag, sizeof(int) * f->elenum_all);
  memcpy(eleall2ele, f->eleall2ele, sizeof(int) * (f->elenum_all + 1));
  memcpy(elecheck_eleall, f->elecheck_eleall, sizeof(int) * f->elenum_all);
  memcpy(elebuf2eleall, f->elebuf2eleall, sizeof(int) * f->elenum_all);
  memcpy(tag2eleall, f->tag2eleall, sizeof(int) * (f->maxtag_all + 1));
}
void orc_fix_get_matrix(const orc_fix *f, double *aaa) { memcpy(aaa, f->aaa_all, sizeof(double) * (size_t)f->elenum_all * f->elenum_all); }
void orc_fix_set_matrix(orc_fix *f, const double *aaa, int runstage) { memcpy(f->aaa_all, aaa, sizeof(double) * (size_t)f->elenum_all * f->elenum_all); f->runstage = runstage; }
void orc_fix_get_vectors(const orc_fix *f, double *bbb_all, double *eleallq, double *elesetq) {
  if (bbb_all) memcpy(bbb_all, f->bbb_all, sizeof(double) * f->elenum_all);
  if (eleallq) memcpy(eleallq, f->eleallq, sizeof(double) * f->elenum_all);
  if (elesetq) memcpy(elesetq, f->elesetq, sizeof(double) * f->elenum_all);
}
void orc_fix_get_trig(const orc_fix *f, double *csk, double *snk) {
  size_t n = (size_t)f->elenum_all * f->ks->kcount_flat;
  memcpy(csk, f->csk, sizeof(double) * n);
  memcpy(snk, f->snk, sizeof(double) * n);
}
/* direct call of the real-space b loop for isolated tests: out[elenum] zeroed then accumulated */
void orc_fix_blist_only(orc_fix *f, double *out_local) {
  int i;
  for (i = 0; i < f->elenum; ++i) out_local[i] = 0.0;
  orc_blist_coul_cal(f, out_local);
}
void orc_fix_alist_only(orc_fix *f, double *out /*[elenum*Ne]*/) {
  memset(out, 0, sizeof(double) * (size_t)f->elenum * f->elenum_all);
  orc_alist_coul_cal(f, out);
}

/* =============================================================================================
 * Multi-rank index bookkeeping restated for nprocs simulated ranks (fix_conp.cpp:468-539):
 * given, per rank, the tags of its owned electrode atoms in local storage order at the FIRST
 * post_neighbor (tags_first) and at the CURRENT one (tags_now), produce the permanent numbering
 * and the gather permutation.  counts_*[r] = electrode atoms on rank r; arrays are rank-major.
 * ===========================================================================================*/
void orc_multirank_maps(int nprocs, const int *counts_first, const int *tags_first, const int *counts_now,
                        const int *tags_now, int maxtag, int *eleall2tag, int *tag2eleall, int *displs_now,
                        int *elebuf2eleall) {
  int r, i, n = 0, pos = 0;
  for (r = 0; r < nprocs; ++r) n += counts_first[r];
  for (i = 0; i <= maxtag; ++i) tag2eleall[i] = n;             /* sentinel :521 */
  for (i = 0; i < n; ++i) eleall2tag[i] = tags_first[i];       /* Allgatherv :523 */
  for (i = 0; i < n; ++i) tag2eleall[eleall2tag[i]] = i;       /* :524 */
  for (r = 0; r < nprocs; ++r) { displs_now[r] = pos; pos += counts_now[r]; }   /* :504-508 */
  for (i = 0; i < pos; ++i) elebuf2eleall[i] = tag2eleall[tags_now[i]];         /* :528-535 */
}

/* =============================================================================================
 * PPPM b-vector (pppm_conp.cpp:109-124 elyte_map_rho_pois, :126-170 elyte_particle_map, :172-228 elyte_make_rho,
 * :230-267 elyte_poisson, :269-316 b_cal, :318-344 aaa_map_rho), one rank, double precision.
 *
 * PARITY UNPINNED at the LAMMPS boundary: the stencil coefficients (PPPM::compute_rho_coeff / compute_rho1d), the
 * influence function (PPPM::compute_gf_ik / gf_denom / compute_gf_denom) and the index conventions (shift, shiftone,
 * OFFSET, nlower, nupper) live in LAMMPS src/KSPACE/pppm.cpp @ 27May2021, which is not under /root/reference.  They
 * are restated here from the published Hockney-Eastwood P3M formulation as LAMMPS implements it; the reference's tests
 * hold no numbers for them, so this part is only cross-checked against the Ewald b vector within the PPPM accuracy.
 * The FFT is a plain O(N n) DFT per axis (small meshes only).
 * ===========================================================================================*/
#define ORC_PPPM_OFFSET 16384
#define ORC_PPPM_MAXORDER 8
#define ORC_EPS_HOC 1.0e-7

typedef struct {
  int nx, ny, nz, order, nlower, nupper, nfft;
  double shift, shiftone, delinv[3], delvolinv, boxlo[3], prd[3], zprd_slab, volume, g_ewald;
  int slabflag;
  double rho_coeff[ORC_PPPM_MAXORDER][ORC_PPPM_MAXORDER]; /* [l][m - (1-order)/2] */
  double gf_b[ORC_PPPM_MAXORDER];
  double *greensfn;
} orc_pppm;

static void orc_pppm_rho_coeff(orc_pppm *p) { /* PPPM::compute_rho_coeff */
  const int order = p->order;
  double a[ORC_PPPM_MAXORDER][2 * ORC_PPPM_MAXORDER + 1]; /* a[l][k + order] */
  int j, k, l, m;
  for (l = 0; l < order; l++) for (k = -order; k <= order; k++) a[l][k + order] = 0.0;
  a[0][order] = 1.0;
  for (j = 1; j < order; j++) {
    for (k = -j; k <= j; k += 2) {
      double s = 0.0;
      for (l = 0; l < j; l++) {
        a[l + 1][k + order] = (a[l][k + 1 + order] - a[l][k - 1 + order]) / (l + 1);
        s += pow(0.5, (double)l + 1) * (a[l][k - 1 + order] + pow(-1.0, (double)l) * a[l][k + 1 + order]) / (l + 1);
      }
      a[0][k + order] = s;
    }
  }
  m = 0;
  for (k = -(order - 1); k < order; k += 2) {
    for (l = 0; l < order; l++) p->rho_coeff[l][m] = a[l][k + order];
    m++;
  }
}

static void orc_pppm_rho1d(const orc_pppm *p, double dx, double *w /*[order]*/) { /* PPPM::compute_rho1d, one axis */
  int k, l;
  for (k = 0; k < p->order; k++) {
    double r = 0.0;
    for (l = p->order - 1; l >= 0; l--) r = p->rho_coeff[l][k] + r * dx;
    w[k] = r;
  }
}

static void orc_pppm_gf_denom_setup(orc_pppm *p) { /* PPPM::compute_gf_denom */
  const int order = p->order;
  int k, l, m;
  long long ifact = 1;
  double gaminv;
  for (l = 1; l < order; l++) p->gf_b[l] = 0.0;
  p->gf_b[0] = 1.0;
  for (m = 1; m < order; m++) {
    for (l = m; l > 0; l--) p->gf_b[l] = 4.0 * (p->gf_b[l] * (l - m) * (l - m - 0.5) - p->gf_b[l - 1] * (l - m - 1) * (l - m - 1));
    p->gf_b[0] = 4.0 * (p->gf_b[0] * (l - m) * (l - m - 0.5));
  }
  for (k = 1; k < 2 * order; k++) ifact *= k;
  gaminv = 1.0 / ifact;
  for (l = 0; l < order; l++) p->gf_b[l] *= gaminv;
}

static double orc_pppm_gf_denom(const orc_pppm *p, double x, double y, double z) {
  double sx = 0, sy = 0, sz = 0, s;
  int l;
  for (l = p->order - 1; l >= 0; l--) { sx = p->gf_b[l] + sx * x; sy = p->gf_b[l] + sy * y; sz = p->gf_b[l] + sz * z; }
  s = sx * sy * sz;
  return s * s;
}

static double orc_powsinxx(double x, int n) {
  double yy;
  if (x == 0.0) return 1.0;
  yy = sin(x) / x;
  return pow(yy, n);
}

static void orc_pppm_gf_ik(orc_pppm *p) { /* PPPM::compute_gf_ik, single rank: the FFT brick is the whole mesh */
  const double xprd = p->prd[0], yprd = p->prd[1], zprd_slab = p->zprd_slab, g = p->g_ewald;
  const double unitkx = 2.0 * ORC_PI / xprd, unitky = 2.0 * ORC_PI / yprd, unitkz = 2.0 * ORC_PI / zprd_slab;
  const int nbx = (int)((g * xprd / (ORC_PI * p->nx)) * pow(-log(ORC_EPS_HOC), 0.25));
  const int nby = (int)((g * yprd / (ORC_PI * p->ny)) * pow(-log(ORC_EPS_HOC), 0.25));
  const int nbz = (int)((g * zprd_slab / (ORC_PI * p->nz)) * pow(-log(ORC_EPS_HOC), 0.25));
  const int twoorder = 2 * p->order;
  int k, l, m, n = 0, ax, ay, az;
  for (m = 0; m < p->nz; m++) {
    const int mper = m - p->nz * (2 * m / p->nz);
    const double snz = pow(sin(0.5 * unitkz * mper * zprd_slab / p->nz), 2);
    for (l = 0; l < p->ny; l++) {
      const int lper = l - p->ny * (2 * l / p->ny);
      const double sny = pow(sin(0.5 * unitky * lper * yprd / p->ny), 2);
      for (k = 0; k < p->nx; k++) {
        const int kper = k - p->nx * (2 * k / p->nx);
        const double snx = pow(sin(0.5 * unitkx * kper * xprd / p->nx), 2);
        const double sqk = pow(unitkx * kper, 2) + pow(unitky * lper, 2) + pow(unitkz * mper, 2);
        if (sqk != 0.0) {
          const double numerator = 12.5663706 / sqk;
          const double denominator = orc_pppm_gf_denom(p, snx, sny, snz);
          double sum1 = 0.0;
          for (ax = -nbx; ax <= nbx; ax++) {
            const double qx = unitkx * (kper + p->nx * ax);
            const double sx = exp(-0.25 * pow(qx / g, 2));
            const double wx = orc_powsinxx(0.5 * qx * xprd / p->nx, twoorder);
            for (ay = -nby; ay <= nby; ay++) {
              const double qy = unitky * (lper + p->ny * ay);
              const double sy = exp(-0.25 * pow(qy / g, 2));
              const double wy = orc_powsinxx(0.5 * qy * yprd / p->ny, twoorder);
              for (az = -nbz; az <= nbz; az++) {
                const double qz = unitkz * (mper + p->nz * az);
                const double sz = exp(-0.25 * pow(qz / g, 2));
                const double wz = orc_powsinxx(0.5 * qz * zprd_slab / p->nz, twoorder);
                const double dot1 = unitkx * kper * qx + unitky * lper * qy + unitkz * mper * qz;
                const double dot2 = qx * qx + qy * qy + qz * qz;
                sum1 += (dot1 / dot2) * sx * sy * sz * wx * wy * wz;
              }
            }
          }
          p->greensfn[n++] = numerator * sum1 / denominator;
        } else p->greensfn[n++] = 0.0;
      }
    }
  }
}

orc_pppm *orc_pppm_create(int nx, int ny, int nz, int order, double g_ewald, double slab_volfactor, int slabflag,
                          const double *boxlo, const double *prd) {
  orc_pppm *p = (orc_pppm *)calloc(1, sizeof(orc_pppm));
  int c;
  p->nx = nx; p->ny = ny; p->nz = nz; p->order = order; p->g_ewald = g_ewald; p->slabflag = slabflag;
  p->nlower = -(order - 1) / 2; p->nupper = order / 2;
  if (order % 2) { p->shift = ORC_PPPM_OFFSET + 0.5; p->shiftone = 0.0; }
  else { p->shift = ORC_PPPM_OFFSET; p->shiftone = 0.5; }
  for (c = 0; c < 3; ++c) { p->boxlo[c] = boxlo[c]; p->prd[c] = prd[c]; }
  p->zprd_slab = prd[2] * slab_volfactor;
  p->volume = prd[0] * prd[1] * p->zprd_slab;
  p->delinv[0] = nx / prd[0]; p->delinv[1] = ny / prd[1]; p->delinv[2] = nz / p->zprd_slab;
  p->delvolinv = p->delinv[0] * p->delinv[1] * p->delinv[2];
  p->nfft = nx * ny * nz;
  p->greensfn = (double *)malloc(sizeof(double) * p->nfft);
  orc_pppm_rho_coeff(p);
  orc_pppm_gf_denom_setup(p);
  orc_pppm_gf_ik(p);
  return p;
}

void orc_pppm_destroy(orc_pppm *p) { if (p) { free(p->greensfn); free(p); } }
void orc_pppm_tables(const orc_pppm *p, double *rho_coeff /*[order*order] l-major*/, double *greensfn) {
  int l, m;
  for (l = 0; l < p->order; ++l) for (m = 0; m < p->order; ++m) rho_coeff[l * p->order + m] = p->rho_coeff[l][m];
  memcpy(greensfn, p->greensfn, sizeof(double) * p->nfft);
}

/* plain DFT along one axis of a [nz][ny][nx] complex array (sign = -1 forward like LAMMPS FFT3d flag 1, +1 backward) */
static void orc_dft_axis(double *re, double *im, int nx, int ny, int nz, int axis, int sign) {
  const int n = axis == 0 ? nx : (axis == 1 ? ny : nz);
  const int stride = axis == 0 ? 1 : (axis == 1 ? nx : nx * ny);
  double *cw = (double *)malloc(sizeof(double) * n), *sw = (double *)malloc(sizeof(double) * n);
  double *tr = (double *)malloc(sizeof(double) * n), *ti = (double *)malloc(sizeof(double) * n);
  int a, b, f, t;
  const int na = axis == 0 ? ny : nx, nb = axis == 2 ? ny : nz;
  for (t = 0; t < n; ++t) { cw[t] = cos(2.0 * ORC_PI * t / n); sw[t] = sign * sin(2.0 * ORC_PI * t / n); }
  for (b = 0; b < nb; ++b)
    for (a = 0; a < na; ++a) {
      size_t base;
      if (axis == 0) base = ((size_t)b * ny + a) * nx;
      else if (axis == 1) base = (size_t)b * ny * nx + a;
      else base = (size_t)b * nx + a;
      for (f = 0; f < n; ++f) {
        double sr = 0, si = 0;
        for (t = 0; t < n; ++t) {
          const int w = (int)(((long long)f * t) % n);
          const double xr = re[base + (size_t)t * stride], xi = im[base + (size_t)t * stride];
          sr += xr * cw[w] - xi * sw[w];
          si += xr * sw[w] + xi * cw[w];
        }
        tr[f] = sr; ti[f] = si;
      }
      for (f = 0; f < n; ++f) { re[base + (size_t)f * stride] = tr[f]; im[base + (size_t)f * stride] = ti[f]; }
    }
  free(cw); free(sw); free(tr); free(ti);
}

static int orc_wrap(int i, int n) { i %= n; return i < 0 ? i + n : i; }

/* pppm_conp.cpp:269-316 b_cal for electrode coordinates xele[ne][3] (eleall order); bbb[ne] overwritten.
 * u_out (optional, [nfft]) receives the mesh potential. */
void orc_pppm_b_cal(const orc_pppm *p, int nlocal, const double *x, const double *q, const int *echeck, int ne,
                    const double *xele, double *bbb, double *u_out) {
  const int order = p->order, nx = p->nx, ny = p->ny, nz = p->nz;
  double *rho = (double *)calloc(p->nfft, sizeof(double)), *im = (double *)calloc(p->nfft, sizeof(double));
  double w[3][ORC_PPPM_MAXORDER];
  double slabcorr = 0.0;
  int i, c, l, m, n;
  const double scaleinv = 1.0 / ((double)nx * ny * nz);
  for (i = 0; i < nlocal; ++i) {   /* elyte_particle_map + elyte_make_rho */
    int g[3];
    double z0;
    if (echeck[i] != 0 || q[i] == 0) continue;
    for (c = 0; c < 3; ++c) {
      const double xs = (x[3 * i + c] - p->boxlo[c]) * p->delinv[c];
      g[c] = (int)(xs + p->shift) - ORC_PPPM_OFFSET;
      orc_pppm_rho1d(p, g[c] + p->shiftone - xs, w[c]);
    }
    z0 = p->delvolinv * q[i];
    for (n = 0; n < order; n++) {
      const int mz = orc_wrap(n + p->nlower + g[2], nz);
      const double y0 = z0 * w[2][n];
      for (m = 0; m < order; m++) {
        const int my = orc_wrap(m + p->nlower + g[1], ny);
        const double x0 = y0 * w[1][m];
        for (l = 0; l < order; l++) {
          const int mx = orc_wrap(l + p->nlower + g[0], nx);
          rho[((size_t)mz * ny + my) * nx + mx] += x0 * w[0][l];
        }
      }
    }
    slabcorr += 4 * q[i] * ORC_PI * x[3 * i + 2] / p->volume;
  }
  for (c = 0; c < 3; ++c) orc_dft_axis(rho, im, nx, ny, nz, c, -1);   /* elyte_poisson */
  for (i = 0; i < p->nfft; ++i) { rho[i] *= scaleinv * p->greensfn[i]; im[i] *= scaleinv * p->greensfn[i]; }
  for (c = 0; c < 3; ++c) orc_dft_axis(rho, im, nx, ny, nz, c, +1);
  if (u_out) memcpy(u_out, rho, sizeof(double) * p->nfft);
  for (i = 0; i < ne; ++i) {        /* aaa_map_rho weights + the stencil gather of b_cal */
    int g[3];
    double bbbtmp = 0;
    for (c = 0; c < 3; ++c) {
      const double xlo = xele[3 * i + c] - p->boxlo[c];
      g[c] = (int)(xlo * p->delinv[c] + p->shift) - ORC_PPPM_OFFSET;
      orc_pppm_rho1d(p, g[c] + p->shiftone - xlo * p->delinv[c], w[c]);
    }
    for (n = 0; n < order; ++n) {
      const int mz = orc_wrap(n + p->nlower + g[2], nz);
      const double z0 = w[2][n];
      for (m = 0; m < order; ++m) {
        const int my = orc_wrap(m + p->nlower + g[1], ny);
        const double y0 = z0 * w[1][m];
        for (l = 0; l < order; ++l) {
          const int mx = orc_wrap(l + p->nlower + g[0], nx);
          const double x0 = y0 * w[0][l];
          bbbtmp -= x0 * rho[((size_t)mz * ny + my) * nx + mx];
        }
      }
    }
    bbb[i] = bbbtmp;
  }
  if (p->slabflag == 1) for (i = 0; i < ne; ++i) bbb[i] -= xele[3 * i + 2] * slabcorr;
  free(rho); free(im);
}

/* =============================================================================================
 * PPPM coupling beyond b and `compute potential/atom` (8f-4), one rank.  PARITY UNPINNED like the section above.
 * ===========================================================================================*/

/* the stencil spread of one atom set: sel == NULL -> every atom; which = 0 electrolyte (echeck == 0), 1 electrode (echeck != 0),
 * 2 both.  q == 0 atoms add nothing.  rho is accumulated (the caller zeroes it). */
static void orc_pppm_spread(const orc_pppm *p, int n, const double *x, const double *q, const int *echeck, int which, double *rho) {
  const int order = p->order, nx = p->nx, ny = p->ny, nz = p->nz;
  double w[3][ORC_PPPM_MAXORDER];
  int i, c, l, m, k;
  for (i = 0; i < n; ++i) {
    int g[3];
    double z0;
    if (which == 0 && echeck[i] != 0) continue;
    if (which == 1 && echeck[i] == 0) continue;
    if (q[i] == 0) continue;
    for (c = 0; c < 3; ++c) {
      const double xs = (x[3 * i + c] - p->boxlo[c]) * p->delinv[c];
      g[c] = (int)(xs + p->shift) - ORC_PPPM_OFFSET;
      orc_pppm_rho1d(p, g[c] + p->shiftone - xs, w[c]);
    }
    z0 = p->delvolinv * q[i];
    for (k = 0; k < order; k++) {
      const int mz = orc_wrap(k + p->nlower + g[2], nz);
      const double y0 = z0 * w[2][k];
      for (m = 0; m < order; m++) {
        const int my = orc_wrap(m + p->nlower + g[1], ny);
        const double x0 = y0 * w[1][m];
        for (l = 0; l < order; l++) rho[((size_t)mz * ny + my) * nx + orc_wrap(l + p->nlower + g[0], nx)] += x0 * w[0][l];
      }
    }
  }
}

/* PPPMCONP::ele_make_rho (pppm_conp.cpp:385-426) and the make_rho override (:434-450): density_brick = elyte_density_brick +
 * ele_density_brick on the periodic mesh (one rank: ghost planes folded).  Any output may be NULL. */
void orc_pppm_make_rho(const orc_pppm *p, int nlocal, const double *x, const double *q, const int *echeck, double *density,
                       double *ele_density, double *elyte_density) {
  double *e = (double *)calloc(p->nfft, sizeof(double)), *l = (double *)calloc(p->nfft, sizeof(double));
  int i;
  orc_pppm_spread(p, nlocal, x, q, echeck, 1, e);
  orc_pppm_spread(p, nlocal, x, q, echeck, 0, l);
  if (ele_density) memcpy(ele_density, e, sizeof(double) * p->nfft);
  if (elyte_density) memcpy(elyte_density, l, sizeof(double) * p->nfft);
  if (density) for (i = 0; i < p->nfft; ++i) density[i] = l[i] + e[i];
  free(e); free(l);
}

/* u_brick as PPPM::compute leaves it when per-atom energies are tallied (what ComputePotentialAtom requires,
 * compute_potential_atom.cpp:128-130): inverse transform of greensfn / N times the transform of the TOTAL density */
void orc_pppm_u_brick(const orc_pppm *p, int nlocal, const double *x, const double *q, const int *echeck, double *u) {
  double *im = (double *)calloc(p->nfft, sizeof(double));
  const double scaleinv = 1.0 / ((double)p->nx * p->ny * p->nz);
  int i, c;
  memset(u, 0, sizeof(double) * p->nfft);
  orc_pppm_spread(p, nlocal, x, q, echeck, 2, u);
  for (c = 0; c < 3; ++c) orc_dft_axis(u, im, p->nx, p->ny, p->nz, c, -1);
  for (i = 0; i < p->nfft; ++i) { u[i] *= scaleinv * p->greensfn[i]; im[i] *= scaleinv * p->greensfn[i]; }
  for (c = 0; c < 3; ++c) orc_dft_axis(u, im, p->nx, p->ny, p->nz, c, +1);
  free(im);
}

/* the stencil sum of PPPMCONP::compute_group_potential / compute_particle_potential (pppm_conp.cpp:452-534): u = - sum w u_brick */
static double orc_pppm_probe(const orc_pppm *p, const double *xi, const double *u) {
  const int order = p->order;
  double w[3][ORC_PPPM_MAXORDER], acc = 0.0;
  int g[3], c, l, m, n;
  for (c = 0; c < 3; ++c) {
    const double xs = (xi[c] - p->boxlo[c]) * p->delinv[c];
    g[c] = (int)(xs + p->shift) - ORC_PPPM_OFFSET;                       /* part2grid, PPPM::particle_map */
    orc_pppm_rho1d(p, g[c] + p->shiftone - xs, w[c]);
  }
  for (n = 0; n < order; n++) {
    const int mz = orc_wrap(n + p->nlower + g[2], p->nz);
    const double z0 = w[2][n];
    for (m = 0; m < order; m++) {
      const int my = orc_wrap(m + p->nlower + g[1], p->ny);
      const double y0 = z0 * w[1][m];
      for (l = 0; l < order; l++) acc -= (y0 * w[0][l]) * u[((size_t)mz * p->ny + my) * p->nx + orc_wrap(l + p->nlower + g[0], p->nx)];
    }
  }
  return acc;
}

/* :487-534: recv[i] for the selected owned atoms.  particle != 0 adds the + 2 g q_i / sqrt(pi) of compute_particle_potential (:483). */
void orc_pppm_group_potential(const orc_pppm *p, int nlocal, const double *x, const double *q, const int *echeck, const int *sel,
                              int particle, double *recv) {
  double *u = (double *)malloc(sizeof(double) * p->nfft);
  int i;
  orc_pppm_u_brick(p, nlocal, x, q, echeck, u);
  for (i = 0; i < nlocal; ++i) {
    if (!sel[i]) continue;
    recv[i] = orc_pppm_probe(p, x + 3 * i, u);
    if (particle) recv[i] += 2 * p->g_ewald * q[i] / 1.77245385090551602729;
  }
  free(u);
}

/* ComputePotentialAtom::compute_peratom (compute_potential_atom.cpp:120-218) for one rank: pair part (:223-308) over the pair
 * style's half list, k-space part through the provider (:165-175), slab correction (:323-345), scaled by qqr2e / qe2f.
 * sel = mask & groupbit, etasel = eta_check (molecule id is molidL or molidR); potential has nlocal entries (+ nghost with newton). */
void orc_compute_potential_atom(const orc_pppm *p, int nlocal, int nghost, const double *x, const double *q, const int *type,
                                const int *echeck, const int *sel, const int *etasel, int inum, const int *ilist, const int *numneigh,
                                const int *first, const int *neigh, int newton, int ntypes, const double *cutsq, double cut_coul,
                                double g_ewald, double eta, int pairflag, int kspaceflag, int slabflag, int qsumflag, double volume,
                                double evscale, double *potential) {
  const int ntotal = nlocal + (newton ? nghost : 0);
  int i, ii, jj;
  for (i = 0; i < ntotal; ++i) potential[i] = 0.0;
  if (pairflag) {
    double cut_coulsq = cut_coul * cut_coul;
    const double cut_erfc = 5.8 * 5.8 / (g_ewald * g_ewald);
    if (cut_coulsq > cut_erfc) cut_coulsq = cut_erfc;
    for (ii = 0; ii < inum; ++ii) {
      const int a = ilist[ii];
      const int gcib = sel[a];
      for (jj = 0; jj < numneigh[a]; ++jj) {
        const int j = neigh[first[a] + jj] & 0x3FFFFFFF;
        const int gcjb = sel[j];
        if ((gcib || gcjb) && (q[a] != 0 || q[j] != 0) && (newton || gcib || j < nlocal)) {
          const double delx = x[3 * a] - x[3 * j], dely = x[3 * a + 1] - x[3 * j + 1], delz = x[3 * a + 2] - x[3 * j + 2];
          double rsq = delx * delx + dely * dely + delz * delz;
          if (rsq < 1e-10) rsq = 1e-10;
          if (rsq < cutsq[type[a] * (ntypes + 1) + type[j]] && rsq < cut_coulsq) {
            const double r = sqrt(rsq), grij = g_ewald * r;
            double expm2 = exp(-grij * grij), t = 1.0 / (1.0 + 0.3275911 * grij);
            double erfc_ = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429)))) * expm2;
            double dudq = erfc_ / r;
            if (eta != 0. && etasel[a] + etasel[j]) {
              const double etarij = (etasel[a] + etasel[j] == 2) ? eta * r / sqrt(2) : eta * r;
              if (etarij < 5.8) {
                expm2 = exp(-etarij * etarij);
                t = 1.0 / (1.0 + 0.3275911 * etarij);
                erfc_ = t * (0.254829592 + t * (-0.284496736 + t * (1.421413741 + t * (-1.453152027 + t * 1.061405429)))) * expm2;
                dudq -= erfc_ / r;
              }
            }
            if (gcib) potential[a] += q[j] * dudq;
            if (j < nlocal || newton) potential[j] += q[a] * dudq;
          }
        }
      }
    }
  }
  if (kspaceflag) {
    double *u = (double *)malloc(sizeof(double) * p->nfft);
    orc_pppm_u_brick(p, nlocal, x, q, echeck, u);
    for (i = 0; i < nlocal; ++i)
      if (sel[i]) {
        potential[i] -= orc_pppm_probe(p, x + 3 * i, u) + 2 * g_ewald * q[i] / 1.77245385090551602729;
        if (eta != 0. && etasel[i]) potential[i] += eta * q[i] * sqrt(2) / 1.77245385090551602729;
      }
    free(u);
    if (slabflag) {                                                     /* :323-345 */
      double qsum = 0.0, slabcorr = 0.0;
      const double pi2vol = 2 * ORC_PI / volume;
      for (i = 0; i < nlocal; ++i) { slabcorr += 2 * pi2vol * q[i] * x[3 * i + 2]; qsum += q[i]; }
      for (i = 0; i < nlocal; ++i)
        if (sel[i]) {
          potential[i] += x[3 * i + 2] * slabcorr;
          if (qsumflag) potential[i] -= pi2vol * qsum * x[3 * i + 2] * x[3 * i + 2];
        }
    }
  }
  for (i = 0; i < ntotal; ++i) potential[i] *= evscale;
}
