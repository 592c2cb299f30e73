#include "compute_potential_atom_hip.h"

#include <cstring>

#include "pppm_conp_hip.h"
#ifndef CONP_GLUE_MOCK
#include "atom.h"
#include "error.h"
#include "force.h"
#include "kspace.h"
#include "neigh_list.h"
#include "pair.h"
#include "update.h"
#endif

using namespace LAMMPS_NS;

/* argument grammar of compute_potential_atom.cpp:49-93 */
ComputePotentialAtomHip::ComputePotentialAtomHip(LAMMPS *lmp, int narg, char **arg)
    : Compute(lmp, narg, arg), provider(nullptr), nmax(0), molidL(-1), molidR(-1), eta(0.), potential(nullptr) {
  if (narg < 3) error->all(FLERR, "Illegal compute pe/atom command");
  peratom_flag = 1; size_peratom_cols = 0; peatomflag = 1; timeflag = 1; comm_reverse = 1;
  etaflag = false; qsumflag = true;
  if (narg == 3) { pairflag = true; kspaceflag = true; }
  else {
    pairflag = false; kspaceflag = false;
    int iarg = 3;
    while (iarg < narg) {
      if (strcmp(arg[iarg], "pair") == 0) pairflag = true;
      else if (strcmp(arg[iarg], "kspace") == 0) kspaceflag = true;
      else if (strcmp(arg[iarg], "noqsum") == 0) qsumflag = false;
      else if (strcmp(arg[iarg], "eta") == 0) {
        if (narg < iarg + 4) error->all(FLERR, "Insufficient arguments for eta flag");
        etaflag = true;
        eta = utils::numeric(FLERR, arg[++iarg], false, lmp);
        molidL = utils::inumeric(FLERR, arg[++iarg], false, lmp);
        molidR = utils::inumeric(FLERR, arg[++iarg], false, lmp);
      } else error->all(FLERR, "Illegal compute potential/atom command");
      iarg++;
    }
    if (etaflag && !pairflag && !kspaceflag) { pairflag = true; kspaceflag = true; }
  }
}

ComputePotentialAtomHip::~ComputePotentialAtomHip() { delete[] potential; }

void ComputePotentialAtomHip::setup() {                       /* :97-112 */
  provider = dynamic_cast<PPPMConpHip *>(force->kspace);
  if (provider == nullptr)
    error->all(FLERR, "Compute requires a compatible KSpace provider like pppm/conp");   /* here: pppm/conp/hip, which owns the device mesh */
}

void ComputePotentialAtomHip::compute_peratom() {             /* :120-218 */
  invoked_peratom = update->ntimestep;
  if (update->eflag_atom != invoked_peratom) error->all(FLERR, "Per-atom energy was not tallied on needed timestep");
  if (atom->nmax > nmax) {
    delete[] potential;
    nmax = atom->nmax;
    potential = new double[nmax];
    vector_atom = potential;
  }
  const int nlocal = atom->nlocal, nall = nlocal + atom->nghost;
  const int ntotal = nlocal + (force->newton ? atom->nghost : 0);
  sel.resize(nall); etasel.resize(nall); echeck.assign(nall, 0); xflat.resize(3 * (size_t)nall); out.assign(nall, 0.0);
  for (int i = 0; i < nall; ++i) {
    sel[i] = (atom->mask[i] & groupbit) ? 1 : 0;
    etasel[i] = (atom->molecule && (atom->molecule[i] == molidL || atom->molecule[i] == molidR)) ? 1 : 0;   /* eta_check :313-318 */
    if (provider->fixconp) echeck[i] = provider->fixconp->electrode_check(i);
    for (int c = 0; c < 3; ++c) xflat[3 * (size_t)i + c] = atom->x[i][c];
  }
  conp_atoms at;
  at.nlocal = nlocal; at.nghost = atom->nghost; at.x = xflat.data(); at.q = atom->q; at.type = atom->type; at.tag = atom->tag;
  at.echeck = echeck.data();
  /* the pair style's half list (:231: force->pair->list), flattened */
  NeighList *l = force->pair->list;
  first.assign(nall, 0); neigh.clear();
  conp_neighlist pl;
  std::memset(&pl, 0, sizeof(pl));
  if (pairflag && l) {
    for (int ii = 0; ii < l->inum; ++ii) {
      const int i = l->ilist[ii];
      first[i] = (int)neigh.size();
      neigh.insert(neigh.end(), l->firstneigh[i], l->firstneigh[i] + l->numneigh[i]);
    }
    if (neigh.empty()) neigh.push_back(0);
    pl.inum = l->inum; pl.ilist = l->ilist; pl.numneigh = l->numneigh; pl.first = first.data(); pl.neigh = neigh.data();
    pl.nneigh = (int64_t)neigh.size();
  }
  conp_potential_args pa;
  pa.pairflag = pairflag ? 1 : 0;
  pa.kspaceflag = (kspaceflag && force->kspace && force->kspace->compute_flag) ? 1 : 0;
  pa.qsumflag = qsumflag ? 1 : 0;
  pa.eta = eta;
  if (conp_compute_potential_atom(provider->handle(), &at, pairflag ? &pl : nullptr, sel.data(), etaflag ? etasel.data() : nullptr, &pa,
                                  out.data()) != CONP_OK)
    error->all(FLERR, conp_last_error());
  for (int i = 0; i < ntotal; ++i) potential[i] = out[i];
}

double ComputePotentialAtomHip::memory_usage() { return (double)nmax * sizeof(double); }
