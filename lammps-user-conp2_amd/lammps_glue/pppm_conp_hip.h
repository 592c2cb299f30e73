/* PPPMConpHip -- `kspace_style pppm/conp/hip`: the PPPM provider of USER-CONP2 (`KSpaceStyle(pppm/conp,PPPMCONP)`, pppm_conp.h:21)
 * with its fix-side methods served by libconp_hip.so.  It is LAMMPS' PPPM (all of its own compute() stays as it is) AND a
 * KSpaceModule of the reference, so that
 *   - FixConp's `pppm` keyword finds it by `dynamic_cast<KSpaceModule *>(force->kspace)` (fix_conp.cpp:402) and registers itself,
 *   - `compute potential/atom` (compute_potential_atom.cpp:103-111) finds a provider with mesh potentials,
 *   - FixConpHip's `pppm` keyword reads the mesh (nx_pppm, ny_pppm, nz_pppm, order) off it like off any PPPM style.
 * What it serves through the C ABI: b_cal (pppm_conp.cpp:269-316), a_cal (:91-101: the Ewald matrix), compute_particle_potential /
 * compute_group_potential (:452-534), and the density bricks of ele_make_rho / make_rho (:385-450) for a host that wants them. */
#ifdef KSPACE_CLASS

KSpaceStyle(pppm/conp/hip,PPPMConpHip)

#else

#ifndef LMP_PPPM_CONP_HIP_H
#define LMP_PPPM_CONP_HIP_H

#include <vector>

#include "conp_hip.h"
#ifdef CONP_GLUE_MOCK
#include "mock_lammps/conp2_mock.h"
#else
#include "kspacemodule.h"
#include "pppm.h"
#endif

namespace LAMMPS_NS {

class PPPMConpHip : public PPPM, public KSpaceModule {
 public:
  explicit PPPMConpHip(class LAMMPS *);
  ~PPPMConpHip() override;
  void conp_setup(bool lowmem) override;                              /* pppm_conp.h:27 + handle creation */
  void conp_post_neighbor(bool, bool) override;                       /* pppm_conp.cpp:66-89 */
  void a_cal(double *aaa) override;                                   /* :91-101 */
  void a_read() override {}                                           /* :103-107: the handle sizes its own mesh arrays */
  void b_cal(double *bbb) override;                                   /* :269-316 */
  void conp_pre_force() override;                                     /* pppm_conp.h:42 elyte_mapped = false: the library drops the brick it kept */
  void update_charge() override {}                                    /* :45 ele_make_rho: the electrode brick is made when asked for */
  double compute_particle_potential(int i) override;                  /* :452-485 */
  void compute_group_potential(int groupbit, double *recv) override;  /* :487-534 */
  double return_qsum() override { return qsum; }                      /* pppm_conp.h:48 */
  /* PPPM::compute's two virtual steps (pppm_conp.cpp:428-450): once b_cal has run, the density PPPM::compute transforms is the
   * electrolyte brick b_cal made for this step + the electrode brick of the charges update_charge wrote -- no second spread of
   * every atom.  Before the first b_cal (or without a registered fix) the base class's own steps run. */
  void particle_map() override;
  void make_rho() override;
  /* the density the make_rho override (:434-450) hands to PPPM::compute: electrolyte brick + electrode brick, [nz][ny][nx] */
  void total_density(double *density_brick);
  conp_fix *handle() { return h; }

 private:
  conp_fix *h;
  bool first, bcal_done = false;
  std::vector<double> dens;
  std::vector<int> echeck, lib_tag2eleall, nolist, sel;
  std::vector<double> xflat, cutsq0;
  void fail_if(int status);
  conp_atoms view();
};

}  // namespace LAMMPS_NS
#endif
#endif
