// FixConpHip: LAMMPS-facing glue.  Everything numerical happens behind the C ABI of include/conp_hip.h.
#include "fix_conp_hip.h"

#include <cstring>

#ifdef CONP_GLUE_MOCK
#include "mock_lammps/mpi_mock.h"
#else
#include <mpi.h>
#endif

#ifndef CONP_GLUE_MOCK
#include "atom.h"
#include "comm.h"
#include "domain.h"
#include "error.h"
#include "force.h"
#include "group.h"
#include "input.h"
#include "kspace.h"
#include "modify.h"
#include "neigh_list.h"
#include "neigh_request.h"
#include "neighbor.h"
#include "pair.h"
#include "update.h"
#include "variable.h"
#endif

#include "conp_mpi_comm.h"

using namespace LAMMPS_NS;
using namespace FixConst;

// ---- conp_comm on MPI: the collectives FixConp makes on `world` (fix_conp.cpp:415, 492, 523, 535, 643, 822, 1356;
// km_ewald.cpp:77, 784), handed to the library as callbacks.  ctx = &world.
// (conp_mpi_comm.h: shared with KSpaceModuleHip, counts checked against MPI's int range)
using conp_glue::cb_allreduce_sum; using conp_glue::cb_allreduce_max_int; using conp_glue::cb_allgather_int; using conp_glue::cb_allgatherv;

void FixConpHip::fail_if(int status) {
  if (status != CONP_OK) error->all(FLERR, conp_last_error());
}

FixConpHip::FixConpHip(LAMMPS *lmp, int narg, char **arg)
    : Fix(lmp, narg, arg), h(nullptr), potdiffvar(-1), arequest(-1), brequest(-1), alist(nullptr), blist(nullptr),
      coulpair(nullptr), outf(nullptr), postforceflag(false) {
  fail_if(conp_parse_fix_args(narg, arg, atom->ntypes, &args));           // fix_conp.cpp:79-176, same syntax and errors
  jgroup = group->find(args.group2);
  if (jgroup == -1) error->all(FLERR, "Fix conp group ID does not exist");  // :106-107
  jgroupbit = group->bitmask[jgroup];
  outf = fopen(args.logfile, "w");                                         // :119
  scalar_flag = 1; extscalar = 0; global_freq = 1;                         // :177-179
}

FixConpHip::~FixConpHip() {
  conp_fix_destroy(h);
  if (outf) fclose(outf);
}

int FixConpHip::setmask() { return PRE_EXCHANGE | POST_NEIGHBOR | PRE_FORCE | POST_FORCE | END_OF_STEP; }   // :233-241 (+ PRE_EXCHANGE: un-pinning)

void FixConpHip::init() {
  coulpair = force->pair_match("coul", 0);                                 // :252-258
  if (coulpair == nullptr) coulpair = force->pair_match("coul", 0, 1);
  if (coulpair == nullptr) error->all(FLERR, "Fix conp couldn't detect a Coulombic pair style");
  if (args.potdiff_is_variable) {                                          // :266-272
    potdiffvar = input->variable->find(args.potdiff_var);
    if (potdiffvar < 0) error->all(FLERR, "Fix conp potential difference variable does not exist");
    if (!input->variable->equalstyle(potdiffvar)) error->all(FLERR, "Fix conp potential difference variable is invalid style");
  }
  if (alist == nullptr || blist == nullptr) {                              // :279-291
    if (args.smartlist) request_smartlist();
    else {
      int irequest = neighbor->request(this, instance_me);
      neighbor->requests[irequest]->pair = 0;
      neighbor->requests[irequest]->fix = 1;
    }
  }
  if (h == nullptr) {
    const int nt1 = atom->ntypes + 1;
    cutsq_flat.resize((size_t)nt1 * nt1);
    for (int i = 0; i < nt1; ++i)
      for (int j = 0; j < nt1; ++j) cutsq_flat[(size_t)i * nt1 + j] = (i && j) ? coulpair->cutsq[i][j] : 0.0;
    int itmp;
    conp_env env;
    std::memset(&env, 0, sizeof(env));
    env.qqrd2e = force->qqrd2e; env.qqr2e = force->qqr2e; env.qe2f = force->qe2f; env.dielectric = force->dielectric;
    env.newton_pair = force->newton_pair;
    env.g_ewald = force->kspace->g_ewald; env.accuracy = force->kspace->accuracy;
    env.slab_volfactor = force->kspace->slab_volfactor; env.slabflag = force->kspace->slabflag;
    env.xprd = domain->xprd; env.yprd = domain->yprd; env.zprd = domain->zprd; env.boxlo_z = domain->boxlo[2];
    env.boxlo_x = domain->boxlo[0]; env.boxlo_y = domain->boxlo[1];
    // `pppm` keyword (fix_conp.cpp:400-405 looks for a pppm/conp kspace style): the mesh and stencil order LAMMPS' PPPM chose are
    // public KSpace members; with a non-mesh style they stay 0 and the library answers with the reference's error message
    if (args.pppm) {
      env.pppm_nx = force->kspace->nx_pppm; env.pppm_ny = force->kspace->ny_pppm; env.pppm_nz = force->kspace->nz_pppm;
      env.pppm_order = force->kspace->order;
    }
    env.ntypes = atom->ntypes; env.cutsq = cutsq_flat.data();
    env.cut_coul = *(double *)coulpair->extract("cut_coul", itmp);
    env.one_electrode = (groupbit == jgroupbit);                           // :295
    // several MPI ranks (spatial decomposition): every rank drives its own handle on its own atoms and lists; device -1 lets the
    // library take GPU (rank mod visible devices), so that ranks of a node spread over its GPUs or share one
    env.device = comm->nprocs > 1 ? -(2 + conp_glue::node_local_rank(world)) : 0;     // ranks of a NODE spread over its GPUs
    env.rank = comm->me; env.nranks = comm->nprocs;
    // ghosts that are periodic images of owned atoms are rebuilt on the device; a rank whose ghosts belong to other ranks fails
    // the library's check at post_neighbor and uploads them as they are
    env.ghost_images = 1;
    fail_if(conp_fix_create(&args, &env, &h));
    if (comm->nprocs > 1) {
      conp_comm cc;
      cc.ctx = &world; cc.rank = comm->me; cc.nranks = comm->nprocs;
      cc.allreduce_sum = cb_allreduce_sum; cc.allreduce_max_int = cb_allreduce_max_int;
      cc.allgather_int = cb_allgather_int; cc.allgatherv = cb_allgatherv;
      fail_if(conp_fix_set_comm(h, &cc));
    }
    for (auto &toks : pending_modify) {
      std::vector<const char *> ptrs;
      for (auto &tk : toks) ptrs.push_back(tk.c_str());
      int n = 0;
      fail_if(conp_fix_modify_param(h, (int)ptrs.size(), ptrs.data(), &n));
    }
    pending_modify.clear();
  }
}

// etypes skip lists, same requests as fix_conp.cpp:304-361
void FixConpHip::request_smartlist() {
  const int ntypes = atom->ntypes;
  int *iskip_a = new int[ntypes + 1], *iskip_b = new int[ntypes + 1];
  int **ijskip_a = new int *[ntypes + 1], **ijskip_b = new int *[ntypes + 1];
  for (int i = 0; i <= ntypes; ++i) {
    ijskip_a[i] = new int[ntypes + 1]; ijskip_b[i] = new int[ntypes + 1];
    iskip_a[i] = 1; iskip_b[i] = 0;
    for (int j = 0; j <= ntypes; ++j) ijskip_a[i][j] = 1;
  }
  for (int e = 0; e < args.eletypenum; ++e) { iskip_a[args.eletypes[e]] = 0; ijskip_a[args.eletypes[e]][args.eletypes[e]] = 0; }
  for (int i = 0; i <= ntypes; ++i)
    for (int j = 0; j <= ntypes; ++j) ijskip_b[i][j] = ((!!iskip_a[i]) ^ (!!iskip_a[j])) ? 0 : 1;
  if (args.a_matrix_f == 0) {
    arequest = neighbor->request(this, instance_me);
    NeighRequest *r = neighbor->requests[arequest];
    r->pair = 0; r->fix = 1; r->half = 1; r->full = 0; r->occasional = 1; r->skip = 1; r->iskip = iskip_a; r->ijskip = ijskip_a;
  }
  brequest = neighbor->request(this, instance_me);
  NeighRequest *r = neighbor->requests[brequest];
  r->pair = 0; r->fix = 1; r->half = 1; r->full = 0; r->skip = 1; r->iskip = iskip_b; r->ijskip = ijskip_b;
}

void FixConpHip::init_list(int, NeighList *ptr) {                         // :365-378
  if (args.smartlist) {
    if (ptr->index == arequest) alist = ptr;
    else if (ptr->index == brequest) blist = ptr;
  } else { alist = ptr; blist = ptr; }
}

conp_atoms FixConpHip::view() {
  const int nall = atom->nlocal + atom->nghost;
  echeck.resize(nall);
  for (int i = 0; i < nall; ++i)
    echeck[i] = (atom->mask[i] & groupbit) ? 1 : ((atom->mask[i] & jgroupbit) ? -1 : 0);   // electrode_check :599-605
  conp_atoms a;
  // atom->x is a LAMMPS 2-d array (Memory::create): one contiguous [nmax][3] block behind the row pointers
  a.nlocal = atom->nlocal; a.nghost = atom->nghost; a.x = nall ? &atom->x[0][0] : nullptr; a.q = atom->q; a.type = atom->type;
  a.tag = atom->tag;
  a.echeck = echeck.data();
  return a;
}

void FixConpHip::PinnedInts::reserve(size_t want) {
  if (want <= cap) return;
  const size_t ncap = want + want / 2 + 64;
  int *np_ = static_cast<int *>(conp_host_alloc(ncap * sizeof(int)));
  if (!np_) throw std::bad_alloc();
  if (n) std::memcpy(np_, p, n * sizeof(int));
  conp_host_free(p);
  p = np_; cap = ncap;
}

// atom->x / atom->q page-locked in place between re-neighbourings (include/conp_hip.h "page-locked host memory"): the per-step
// upload of pre_force is then an asynchronous DMA transfer straight out of LAMMPS' arrays.  They move only when the per-atom arrays
// grow (Atom::avec->grow), which happens inside a re-neighbouring step after pre_exchange: un-pinned there, pinned again in
// post_neighbor with whatever the arrays are then.  A refusal leaves the staged copy in use -- not an error.
void FixConpHip::pin_atoms() {
  const int nall = atom->nlocal + atom->nghost;
  const double *x = nall ? &atom->x[0][0] : nullptr;
  const int n = atom->nmax > nall ? atom->nmax : nall;
  if (x == pinned_x && atom->q == pinned_q && n == pinned_n) return;
  pinned_x = pinned_q = nullptr; pinned_n = 0;
  if (conp_fix_pin_host_arrays(h, x, atom->q, n) == CONP_OK) { pinned_x = x; pinned_q = atom->q; pinned_n = n; }
}

void FixConpHip::pre_exchange() {
  if (h && pinned_x) { (void)conp_fix_unpin_host_arrays(h); pinned_x = pinned_q = nullptr; pinned_n = 0; }
}

// LAMMPS pages firstneigh; the library wants it flat (once per re-neighbour, not per step)
void FixConpHip::push_list(int which, NeighList *l, PinnedInts &first, PinnedInts &neigh) {
  const int nall = atom->nlocal + atom->nghost;
  first.assign(nall, 0);
  neigh.clear();
  for (int ii = 0; ii < l->inum; ++ii) {
    const int i = l->ilist[ii];
    first.data()[i] = (int)neigh.size();
    neigh.append(l->firstneigh[i], l->firstneigh[i] + l->numneigh[i]);
  }
  if (neigh.empty()) neigh.push_back(0);
  conp_neighlist v;
  v.inum = l->inum; v.ilist = l->ilist; v.numneigh = l->numneigh; v.first = first.data(); v.neigh = neigh.data();
  v.nneigh = (int64_t)neigh.size();
  fail_if(conp_fix_init_list(h, which, &v));
}

double FixConpHip::potdiff_now() {                                         // :1143
  return args.potdiff_is_variable ? input->variable->compute_equal(potdiffvar) : args.potdiff;
}

void FixConpHip::setup_post_neighbor() {                                   // :382-385
  if (alist->occasional) { neighbor->build(0); neighbor->build_one(alist, 1); }   // :1212-1215 (a_cal needs it once)
  if (alist == blist) push_list(2, blist, first_b, neigh_b);
  else { push_list(0, alist, first_a, neigh_a); push_list(1, blist, first_b, neigh_b); }
  conp_atoms a = view();
  fail_if(conp_fix_setup_post_neighbor(h, &a));
  pin_atoms();
}

void FixConpHip::setup_pre_force(int) {                                    // :387-391
  force->kspace->setup();
  conp_atoms a = view();
  fail_if(conp_fix_setup_pre_force(h, &a, (int64_t)update->ntimestep, potdiff_now()));
  flush_log();                            // "A matrix calculating ..." / "A matrix calculation time" (:787, :857), CG lines
}

void FixConpHip::post_neighbor() {                                         // :468-539
  push_list(alist == blist ? 2 : 1, blist, first_b, neigh_b);
  conp_atoms a = view();
  fail_if(conp_fix_post_neighbor(h, &a));
  pin_atoms();
}

void FixConpHip::pre_force(int) {                                          // :543-573
  conp_atoms a = view();
  fail_if(conp_fix_pre_force(h, &a, (int64_t)update->ntimestep, potdiff_now()));
  if (update->ntimestep % args.everynum == 0 && update->laststep == update->ntimestep)    // :553-568
    fail_if(conp_fix_write_timing(h));
  flush_log();
}

// the reference prints to `outf` from rank 0 only (me == 0 guards at :563, :786, :856, :920)
void FixConpHip::flush_log() {
  const char *text = conp_fix_log_drain(h);
  if (outf && comm->me == 0 && text[0]) { fputs(text, outf); fflush(outf); }
  const char *mesg = conp_fix_mesg_drain(h);                      // "conp output: <e,e> / <d,d>" -> screen + log.lammps
  if (comm->me == 0 && mesg[0]) utils::logmesg(lmp, mesg);
}

// fix_conp.cpp:577-588 post_force / end_of_step -> force_cal (:1163-1201)
void FixConpHip::post_force(int) {
  postforceflag = true;
  conp_atoms a = view();
  const int nall = atom->nlocal + atom->nghost;
  double ek = 0.0, ec = 0.0, vir[6];
  // atom->f is a contiguous [nmax][3] block like atom->x: the library accumulates into it directly (and touches it only when
  // some pair is inside the Gaussian overlap range); x, q of this step are already on the device
  fail_if(conp_fix_post_force_step(h, &a, (int64_t)update->ntimestep, nall ? &atom->f[0][0] : nullptr, &ek, &ec, vir));
  if (force->kspace->energy) force->kspace->energy += ek;       // :1165 (the reference adds only when kspace tallied energy)
  force->pair->eng_coul += ec;                                   // what Pair::ev_tally accumulates (:1436)
  for (int k = 0; k < 6; ++k) force->pair->virial[k] += vir[k];
}

void FixConpHip::end_of_step() {
  if (!postforceflag) post_force(0);
  postforceflag = false;
}

// fix_conp.cpp:1482-1515: fix_modify ID ehgo kappa X | ehgo coeff types eta u0|auto
// fix_modify arrives before init(), i.e. before the library handle exists (the handle needs g_ewald, which kspace only
// knows at init): remember the tokens, replay them in init()
int FixConpHip::modify_param(int narg, char **arg) {
  if (narg < 1 || std::strcmp(arg[0], "ehgo") != 0) return 0;
  if (!args.ehgo) error->all(FLERR, "Can't fix_modify conp parameters in basic pair mode");
  int used = 0;
  if (narg >= 2 && std::strcmp(arg[1], "kappa") == 0) used = 3;
  else if (narg >= 2 && std::strcmp(arg[1], "coeff") == 0) used = 5;
  else error->all(FLERR, "Invalid entry for EHGO coeff setting");
  if (narg != used) error->all(FLERR, "Invalid number of inputs for EHGO coeff setting");
  if (h) { int n = 0; fail_if(conp_fix_modify_param(h, narg, arg, &n)); }
  else pending_modify.emplace_back(arg, arg + narg);
  return used;
}

double FixConpHip::compute_scalar() { return conp_fix_compute_scalar(h); }  // :592-595
