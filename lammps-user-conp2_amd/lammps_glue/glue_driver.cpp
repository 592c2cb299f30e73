// glue_driver -- a miniature host that drives FixConpHip the way LAMMPS does, on top of the interface mock.
// It exists to EXECUTE the C++ glue (constructor / init / init_list / setup_* / post_neighbor / pre_force / post_force /
// compute_scalar / modify_param, log-file writing, error->all propagation, and the MPI-backed conp_comm callbacks) against
// libconp_hip.so on a real GPU; the numerical content is checked by tests/test_gpu_glue.py against the ctypes path through the
// same library.
//
// usage: glue_driver CASEFILE                      one rank, FixConpHip (INTEGRATION.md mode A)
//        glue_driver CASEFILE provider             mode B: the reference's FixConp keeps its loops, only `kspmod` is replaced:
//                                                  a stand-in for the fix's public members is registered with KSpaceModuleHip
//                                                  (register_fix, fix_conp.cpp:409) and conp_setup / conp_post_neighbor / a_cal /
//                                                  b_cal run; output "a I J VALUE" (local row I, global column J), "b I VALUE"
//        glue_driver ranks N CASE_0 .. CASE_N-1 [provider]
//                                                  N rank THREADS, one per sub-domain case file (LAMMPS' spatial decomposition:
//                                                  owned atoms + ghosts + lists of that rank), collectives through the MPI mock;
//                                                  every output line is prefixed "rank R "
// case file (whitespace-separated tokens, written by the test):
//   ntypes nlocal nghost natoms
//   xprd yprd zprd boxlo_x boxlo_y boxlo_z
//   g_ewald accuracy slab_volfactor slabflag nx_pppm ny_pppm nz_pppm order      (mesh 0 0 0 0 = not a PPPM style)
//   qqrd2e qqr2e qe2f dielectric newton_pair cut_coul
//   cutsq[(ntypes+1)^2]
//   ngroups  { name bit } ...
//   per atom (nlocal + nghost): tag type mask q x y z
//   nlists   { inum  { i numneigh j... } ... } ...      list 0 = only / ele-ele list, list 1 = ele-electrolyte list
//   variable: name value   ("-" = none)
//   nmodify { ntok tok... } ...                          fix_modify lines
//   narg tok...                                          the fix command
//   nsteps { ntimestep potdiff_value reneighbor(0/1) has_x(0/1) [x y z per atom] } ...
// output (stdout): "scalar STEP VALUE", "q STEP TAG VALUE" for owned electrode atoms, "f STEP ..." sums, "ERROR: msg" + exit 2.
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <algorithm>
#include <vector>

#define CONP_GLUE_MOCK 1
#include "fix_conp_hip.h"
#include "kspacemodule_hip.h"
#include "pppm_conp_hip.h"

using namespace LAMMPS_NS;

namespace {

struct Out {
  std::string text;
  void f(const char *fmt, ...) {
    char line[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    text += line;
  }
};

// one rank: its own mock LAMMPS instance, atoms and lists from its case file
int run_case(const char *path, bool provider, int me, int nprocs, MockCommRank *world, Out &out) {
  std::ifstream in(path);
  if (!in) { out.f("ERROR: cannot open %s\n", path); return 1; }

  LAMMPS lmp{};
  Memory memory; Error error; Atom atom{}; Force force{}; Domain domain{}; Update update{}; Comm comm{}; Group group{};
  Variable variable; Input input{}; Neighbor neighbor{}; Modify modify; KSpace kspace{}; Pair pair{};
  lmp.world = world;
  lmp.memory = &memory; lmp.error = &error; lmp.atom = &atom; lmp.force = &force; lmp.domain = &domain; lmp.update = &update;
  lmp.comm = &comm; lmp.group = &group; lmp.input = &input; lmp.neighbor = &neighbor; lmp.modify = &modify;
  lmp.screen = me == 0 ? stdout : nullptr; lmp.logfile = nullptr;
  input.variable = &variable;
  force.kspace = &kspace; force.pair = &pair;
  comm.me = me; comm.nprocs = nprocs;
  char style[] = "verlet";
  update.integrate_style = style;

  int ntypes, nlocal, nghost;
  long natoms;
  in >> ntypes >> nlocal >> nghost >> natoms;
  const int nall = nlocal + nghost;
  in >> domain.xprd >> domain.yprd >> domain.zprd >> domain.boxlo[0] >> domain.boxlo[1] >> domain.boxlo[2];
  domain.zprd_half = 0.5 * domain.zprd;
  in >> kspace.g_ewald >> kspace.accuracy >> kspace.slab_volfactor >> kspace.slabflag >> kspace.nx_pppm >> kspace.ny_pppm >>
      kspace.nz_pppm >> kspace.order;
  kspace.energy = 0.0;
  in >> force.qqrd2e >> force.qqr2e >> force.qe2f >> force.dielectric >> force.newton_pair >> pair.cut_coul;
  const int nt1 = ntypes + 1;
  std::vector<double> cutsq_store((size_t)nt1 * nt1);
  std::vector<double *> cutsq_rows(nt1);
  for (auto &v : cutsq_store) in >> v;
  for (int i = 0; i < nt1; ++i) cutsq_rows[i] = cutsq_store.data() + (size_t)i * nt1;
  pair.cutsq = cutsq_rows.data();
  pair.eng_coul = 0.0;
  for (double &v : pair.virial) v = 0.0;

  int ngroups;
  in >> ngroups;
  std::vector<int> bitmask(32, 0);
  group.bitmask = bitmask.data();
  group.ngroup = ngroups;
  for (int g = 0; g < ngroups; ++g) in >> group.names[g] >> bitmask[g];

  std::vector<int> tag(nall), type(nall), mask(nall);
  std::vector<double> q(nall), xs((size_t)nall * 3), fs((size_t)nall * 3, 0.0);
  std::vector<double *> xrows(nall), frows(nall);
  int maxtag = 0;
  for (int i = 0; i < nall; ++i) {
    in >> tag[i] >> type[i] >> mask[i] >> q[i] >> xs[3 * (size_t)i] >> xs[3 * (size_t)i + 1] >> xs[3 * (size_t)i + 2];
    xrows[i] = &xs[3 * (size_t)i]; frows[i] = &fs[3 * (size_t)i];
    if (tag[i] > maxtag) maxtag = tag[i];
  }
  std::vector<int> map_array(maxtag + 1, -1);
  for (int i = nall - 1; i >= 0; --i) map_array[tag[i]] = i;      // owned atoms win over their ghosts, like Atom::map
  atom.nlocal = nlocal; atom.nghost = nghost; atom.ntypes = ntypes; atom.natoms = natoms;
  atom.x = xrows.data(); atom.f = frows.data(); atom.q = q.data(); atom.type = type.data(); atom.mask = mask.data();
  atom.tag = tag.data(); atom.map_array = map_array.data(); atom.map_size = maxtag + 1;

  int nlists;
  in >> nlists;
  struct ListStore { std::vector<int> ilist, numneigh, neigh; std::vector<size_t> first; std::vector<int *> firstneigh; NeighList nl{}; };
  std::vector<ListStore> lists(nlists);
  for (auto &L : lists) {
    int inum;
    in >> inum;
    L.numneigh.assign(nall, 0); L.first.assign(nall, 0);
    for (int ii = 0; ii < inum; ++ii) {
      int i, n;
      in >> i >> n;
      L.ilist.push_back(i); L.numneigh[i] = n; L.first[i] = L.neigh.size();
      for (int k = 0; k < n; ++k) { int j; in >> j; L.neigh.push_back(j); }
    }
    if (L.neigh.empty()) L.neigh.push_back(0);
    L.firstneigh.assign(nall, L.neigh.data());
    for (int i = 0; i < nall; ++i) L.firstneigh[i] = L.neigh.data() + L.first[i];
    L.nl.inum = inum; L.nl.ilist = L.ilist.data(); L.nl.numneigh = L.numneigh.data(); L.nl.firstneigh = L.firstneigh.data();
  }

  std::string vname;
  double vvalue;
  in >> vname >> vvalue;
  if (vname != "-") { variable.name = vname; variable.value = vvalue; }

  int nmodify;
  in >> nmodify;
  std::vector<std::vector<std::string>> modify_lines(nmodify);
  for (auto &ml : modify_lines) { int n; in >> n; ml.resize(n); for (auto &t : ml) in >> t; }

  int narg;
  in >> narg;
  std::vector<std::string> toks(narg);
  for (auto &t : toks) in >> t;
  std::vector<char *> fargv;
  for (auto &t : toks) fargv.push_back(const_cast<char *>(t.c_str()));

  std::vector<NeighRequest *> requests(8, nullptr);
  neighbor.requests = requests.data();

  if (provider) {
    // Mode B.  What FixConp::linalg_init / post_neighbor do around the provider (fix_conp.cpp:401-416, 468-539), with a stand-in
    // that carries the fix's PUBLIC members: the index maps are made the way the fix makes them (one more handle serves as the
    // fix's bookkeeping here; in a LAMMPS tree the reference's own FixConp fills them).
    auto must = [&](int rc) { if (rc != CONP_OK) throw std::runtime_error(conp_last_error()); };
    conp_fix_args fa;
    std::vector<const char *> cargv;
    for (auto &t : toks) cargv.push_back(t.c_str());
    must(conp_parse_fix_args(narg, cargv.data(), ntypes, &fa));
    const int gb = bitmask[group.find(toks[1])], jgb = bitmask[group.find(toks[4])];
    FixConp fixstub;
    fixstub.eta = fa.eta;
    fixstub.check = [&](int i) { return (mask[i] & gb) ? 1 : ((mask[i] & jgb) ? -1 : 0); };
    // the fix's own maps: FixConp::post_neighbor's algorithm (conp_host_index restates it bit-exactly; one rank here -- in the
    // `ranks` mode the stand-in's numbering comes from a bookkeeping handle that talks to the other ranks)
    conp_env env;
    std::memset(&env, 0, sizeof(env));
    env.qqrd2e = force.qqrd2e; env.qqr2e = force.qqr2e; env.qe2f = force.qe2f; env.dielectric = force.dielectric;
    env.newton_pair = force.newton_pair; env.g_ewald = kspace.g_ewald; env.accuracy = kspace.accuracy;
    env.slab_volfactor = kspace.slab_volfactor; env.slabflag = kspace.slabflag;
    env.xprd = domain.xprd; env.yprd = domain.yprd; env.zprd = domain.zprd;
    env.boxlo_x = domain.boxlo[0]; env.boxlo_y = domain.boxlo[1]; env.boxlo_z = domain.boxlo[2];
    env.ntypes = ntypes; env.cutsq = cutsq_store.data(); env.cut_coul = pair.cut_coul; env.nranks = nprocs; env.rank = me;
    env.pppm_nx = kspace.nx_pppm; env.pppm_ny = kspace.ny_pppm; env.pppm_nz = kspace.nz_pppm; env.pppm_order = kspace.order;
    conp_fix *book = nullptr;
    must(conp_fix_create(&fa, &env, &book));
    // (the bookkeeping handle of a multi-rank run would need the comm too; the provider test with ranks uses FixConpHip instead)
    std::vector<std::vector<int>> firsts(nlists);
    for (int l = 0; l < nlists; ++l) {
      ListStore &L = lists[l];
      firsts[l].assign(L.first.begin(), L.first.end());
      conp_neighlist v;
      v.inum = L.nl.inum; v.ilist = L.ilist.data(); v.numneigh = L.numneigh.data(); v.first = firsts[l].data();
      v.neigh = L.neigh.data(); v.nneigh = (int64_t)L.neigh.size();
      must(conp_fix_init_list(book, nlists == 1 ? 2 : l, &v));
    }
    std::vector<int> ec(nall);
    for (int i = 0; i < nall; ++i) ec[i] = fixstub.check(i);
    conp_atoms a;
    a.nlocal = nlocal; a.nghost = nghost; a.x = xs.data(); a.q = q.data(); a.type = type.data(); a.tag = tag.data();
    a.echeck = ec.data();
    must(conp_fix_setup_post_neighbor(book, &a));
    conp_info info;
    must(conp_fix_info(book, &info));
    std::vector<int> ele2tag(info.elenum), ele2eleall(info.elenum), eleall2tag(info.elenum_all), eleall2ele(info.elenum_all + 1),
        echk(info.elenum_all), ebuf(info.elenum_all), tag2eleall(info.maxtag_all + 1);
    must(conp_fix_get_maps(book, ele2tag.data(), ele2eleall.data(), eleall2tag.data(), eleall2ele.data(), echk.data(), ebuf.data(),
                           tag2eleall.data()));
    conp_fix_destroy(book);
    fixstub.elenum = info.elenum; fixstub.elenum_all = info.elenum_all; fixstub.elytenum = info.elytenum;
    fixstub.ele2tag = ele2tag.data(); fixstub.ele2eleall = ele2eleall.data(); fixstub.eleall2tag = eleall2tag.data();
    fixstub.tag2eleall = tag2eleall.data();
    // fix_conp.cpp:401-410: with the `pppm` keyword the provider IS the kspace style, found by dynamic_cast; else
    // kspmod = new ...(lmp); then kspmod->register_fix(this); kspmod->conp_setup(lowmemflag);
    KSpaceModule *kspmod = nullptr;
    PPPMConpHip *pppm_style = nullptr;
    if (fa.pppm) {
      pppm_style = new PPPMConpHip(&lmp);                       // what `kspace_style pppm/conp/hip` creates
      pppm_style->g_ewald = kspace.g_ewald; pppm_style->accuracy = kspace.accuracy; pppm_style->slab_volfactor = kspace.slab_volfactor;
      pppm_style->slabflag = kspace.slabflag; pppm_style->energy = 0.0;
      pppm_style->nx_pppm = kspace.nx_pppm; pppm_style->ny_pppm = kspace.ny_pppm; pppm_style->nz_pppm = kspace.nz_pppm;
      pppm_style->order = kspace.order;
      force.kspace = pppm_style;
      kspmod = dynamic_cast<KSpaceModule *>(force.kspace);      // fix_conp.cpp:402
      if (kspmod == nullptr) throw std::runtime_error("Fix conp couldn't detect a pppm/conp kspace style (which is required with the pppm flag)");
    } else kspmod = new KSpaceModuleHip(&lmp);
    fixstub.kspmod = kspmod;
    kspmod->register_fix(&fixstub);
    kspmod->conp_setup(true);
    kspmod->conp_post_neighbor(true, true);        // fix_conp.cpp:538
    kspmod->a_read();                              // km_ewald.cpp:147-151 a_cal calls a_read first
    std::vector<double> aaa((size_t)info.elenum * info.elenum_all, 0.0), bbb(info.elenum, 0.0);
    kspmod->a_cal(aaa.data());
    kspmod->b_cal(bbb.data());
    for (int i = 0; i < info.elenum; ++i) {
      out.f("b %d %.17g\n", i, bbb[i]);
      for (int j = 0; j < info.elenum_all; ++j) out.f("a %d %d %.17g\n", i, j, aaa[(size_t)i * info.elenum_all + j]);
    }
    for (int i = 0; i < info.elenum; ++i) out.f("m %d %d\n", i, ele2eleall[i]);
    if (pppm_style) {
      // the step goes on: FixConp::update_charge has written the electrode charges, then LAMMPS calls force->kspace->compute() --
      // PPPM::compute's particle_map() / make_rho() are the provider's overrides (pppm_conp.cpp:428-450).  What the base class is
      // handed, how often the base's own steps ran, and whether the library spread the electrolyte a second time:
      conp_info i0, i1;
      must(conp_fix_info(pppm_style->handle(), &i0));
      pppm_style->compute(1, 0);
      must(conp_fix_info(pppm_style->handle(), &i1));
      out.f("rho_calls %d %d spreads %d %d\n", pppm_style->base_particle_map_calls, pppm_style->base_make_rho_calls, i0.pppm_elyte_spreads,
            i1.pppm_elyte_spreads);
      if (const char *rp = std::getenv("GLUE_RHO_OUT")) {
        std::ofstream rf(rp, std::ios::binary);
        rf.write(reinterpret_cast<const char *>(pppm_style->density_fft.data()), (std::streamsize)(pppm_style->density_fft.size() * sizeof(double)));
      }
      kspmod->conp_pre_force();                      // next step (pppm_conp.h:42): the kept brick is dropped ...
      std::vector<double> bbb2(info.elenum, 0.0);
      kspmod->b_cal(bbb2.data());                    // ... and b_cal makes this step's
      double dmax = 0.0;
      for (int i = 0; i < info.elenum; ++i) dmax = std::max(dmax, std::abs(bbb2[i] - bbb[i]));
      out.f("b_again %.17g\n", dmax);
    }
    if (pppm_style) {      // what ComputePotentialAtom asks of the provider (compute_potential_atom.cpp:165-175), group 1 = eleleft
      std::vector<double> recv(nlocal, 0.0);
      kspmod->compute_group_potential(gb, recv.data());
      for (int i = 0; i < nlocal; ++i) if (mask[i] & gb) out.f("u %d %.17g\n", tag[i], recv[i]);
      int i0 = 0;
      while (i0 < nlocal && !(mask[i0] & gb)) ++i0;
      out.f("up %d %.17g\n", tag[i0], kspmod->compute_particle_potential(i0));
      force.kspace = &kspace;
      delete pppm_style;                           // LAMMPS deletes its kspace style, not the fix (fix_conp.cpp:207)
    } else delete kspmod;                          // fix_conp.cpp:207: the fix deletes its Ewald provider; the handle goes with it
    return 0;
  }

  FixConpHip fix(&lmp, narg, fargv.data());
  for (auto &ml : modify_lines) {                    // fix_modify arrives before init(), as in an input script
    std::vector<char *> margv;
    for (auto &t : ml) margv.push_back(const_cast<char *>(t.c_str()));
    const int used = fix.modify_param((int)margv.size(), margv.data());
    if (used == 0) throw std::runtime_error("Illegal fix_modify command");
  }
  fix.init();
  // Neighbor::init hands every request its list (Neighbor::init -> Fix::init_list); an occasional skip list is the
  // ele-ele list (list 0), the perpetual one the ele-electrolyte list (last list)
  for (int r = 0; r < neighbor.nrequest; ++r) {
    const bool occasional = requests[r]->occasional != 0;
    ListStore &L = (neighbor.nrequest == 1 || occasional) ? lists.front() : lists.back();
    L.nl.index = r; L.nl.occasional = requests[r]->occasional;
    fix.init_list(0, &L.nl);
  }
  update.ntimestep = 0;
  int nsteps;
  in >> nsteps;
  // Verlet::setup order: modify->setup_post_neighbor(), then modify->setup_pre_force()
  bool first = true;
  for (int s = 0; s < nsteps; ++s) {
    long ts; double pd; int reneigh, has_x;
    in >> ts >> pd >> reneigh >> has_x;
    if (has_x) for (size_t k = 0; k < xs.size(); ++k) in >> xs[k];
    update.ntimestep = ts; update.laststep = -1;
    if (s == nsteps - 1) update.laststep = ts;
    variable.value = pd;
    if (first) {
      fix.setup_post_neighbor();
      fix.setup_pre_force(0);
      first = false;
    } else {
      if (reneigh) { fix.pre_exchange(); fix.post_neighbor(); }      // Verlet::run: pre_exchange ... neighbor->build ... post_neighbor
      fix.pre_force(0);
    }
    std::fill(fs.begin(), fs.end(), 0.0);
    if (fix.setmask() & FixConst::POST_FORCE) fix.post_force(0);
    fix.end_of_step();
    out.f("scalar %ld %.17g\n", ts, fix.compute_scalar());
    for (int i = 0; i < nlocal; ++i)
      if (mask[i] & (fix.groupbit | bitmask[group.find(toks[4])])) out.f("q %ld %d %.17g\n", ts, tag[i], q[i]);
    double fsum[3] = {0, 0, 0}, fabs_ = 0;
    for (int i = 0; i < nlocal; ++i) for (int c = 0; c < 3; ++c) { fsum[c] += fs[3 * (size_t)i + c]; fabs_ += std::abs(fs[3 * (size_t)i + c]); }
    out.f("f %ld %.17g %.17g %.17g %.17g eng_coul %.17g kspace_energy %.17g\n", ts, fsum[0], fsum[1], fsum[2], fabs_,
          pair.eng_coul, kspace.energy);
  }
  // GLUE_DRIVER_TIME=N: wall time of N more pre_force calls -- the PCIe-inclusive rate a LAMMPS run would see through the glue
  if (const char *tn = std::getenv("GLUE_DRIVER_TIME")) {
    const int n = std::atoi(tn);
    for (int k = 0; k < 3; ++k) { update.ntimestep += 1; fix.pre_force(0); }
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n; ++k) { update.ntimestep += 1; fix.pre_force(0); }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / (n > 0 ? n : 1);
    out.f("time_pre_force_ms %.6f\n", ms);
    const auto t1 = std::chrono::steady_clock::now();
    for (int k = 0; k < n; ++k) { std::fill(fs.begin(), fs.end(), 0.0); fix.post_force(0); }      // same step as the last pre_force
    const double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count() / (n > 0 ? n : 1);
    out.f("time_post_force_ms %.6f\n", ms2);
  }
  return 0;
}

}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: glue_driver CASEFILE [provider] | glue_driver ranks N CASE_0 .. CASE_N-1\n"); return 1; }
  if (std::string(argv[1]) == "ranks") {
    const int n = argc > 2 ? std::atoi(argv[2]) : 0;
    if (n < 1 || argc < 3 + n) { std::fprintf(stderr, "glue_driver ranks N needs N case files\n"); return 1; }
    const bool provider = argc > 3 + n && std::string(argv[3 + n]) == "provider";
    MockWorld world(n);
    std::vector<MockCommRank> ranks(n);
    std::vector<Out> outs(n);
    std::vector<int> rcs(n, 0);
    std::vector<std::thread> threads;
    for (int r = 0; r < n; ++r) {
      ranks[r] = MockCommRank{&world, r};
      threads.emplace_back([&, r] {
        try {
          rcs[r] = run_case(argv[3 + r], provider, r, n, &ranks[r], outs[r]);
        } catch (const std::exception &e) {
          outs[r].f("ERROR: %s\n", e.what());
          rcs[r] = 2;
          world.fail();                      // the other ranks sit in a collective: wake them, they fail too
        }
      });
    }
    for (auto &t : threads) t.join();
    int rc = 0;
    for (int r = 0; r < n; ++r) {
      size_t pos = 0;
      while (pos < outs[r].text.size()) {
        const size_t e = outs[r].text.find('\n', pos);
        const std::string line = outs[r].text.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        std::printf("rank %d %s\n", r, line.c_str());
        if (e == std::string::npos) break;
        pos = e + 1;
      }
      if (rcs[r]) rc = rcs[r];
    }
    return rc;
  }
  Out out;
  int rc = 0;
  try {
    MockCommRank *single = nullptr;
    rc = run_case(argv[1], argc > 2 && std::string(argv[2]) == "provider", 0, 1, single, out);
  } catch (const std::exception &e) {
    out.f("ERROR: %s\n", e.what());
    rc = 2;
  }
  std::fputs(out.text.c_str(), stdout);
  return rc;
}
