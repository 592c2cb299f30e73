/* ComputePotentialAtomHip -- `compute ID group potential/atom/hip [pair] [kspace] [noqsum] [eta ETA molL molR]`: the reference's
 * `compute potential/atom` (compute_potential_atom.h:16, compute_potential_atom.cpp:49-345) with its pair loop, mesh gather and
 * slab correction done by libconp_hip.so in one call (conp_compute_potential_atom).  Same arguments, same errors, same output
 * (per-atom vector in volts).  Needs `kspace_style pppm/conp/hip` when the k-space part is asked for, as the reference needs
 * a "compatible KSpace provider like pppm/conp" (:110). */
#ifdef COMPUTE_CLASS

ComputeStyle(potential/atom/hip,ComputePotentialAtomHip)

#else

#ifndef LMP_COMPUTE_POTENTIAL_ATOM_HIP_H
#define LMP_COMPUTE_POTENTIAL_ATOM_HIP_H

#include <vector>

#include "conp_hip.h"
#ifdef CONP_GLUE_MOCK
#include "mock_lammps/conp2_mock.h"
#else
#include "compute.h"
#endif

namespace LAMMPS_NS {

class ComputePotentialAtomHip : public Compute {
 public:
  ComputePotentialAtomHip(class LAMMPS *, int, char **);
  ~ComputePotentialAtomHip() override;
  void init() override {}
  void setup() override;
  void compute_peratom() override;
  double memory_usage() override;

 private:
  class PPPMConpHip *provider;
  bool pairflag, kspaceflag, etaflag, qsumflag;
  int nmax, molidL, molidR;
  double eta;
  double *potential;
  std::vector<int> sel, etasel, echeck, first, neigh;
  std::vector<double> xflat, out;
};

}  // namespace LAMMPS_NS
#endif
#endif
