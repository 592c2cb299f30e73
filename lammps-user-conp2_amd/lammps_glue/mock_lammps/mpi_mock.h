// INTERFACE MOCK -- NOT MPI.  The handful of MPI calls the glue makes (the same ones FixConp makes on `world`:
// fix_conp.cpp:415, 492, 523, 643, 1356), implemented for rank THREADS inside one process so that glue_driver can execute
// FixConpHip on several "ranks" of a spatially decomposed system without an MPI installation (there is none in the image).
// In a LAMMPS tree the glue includes <mpi.h> instead.
#pragma once
#include <condition_variable>
#include <mutex>
#include <vector>

struct MockWorld {
  int n = 1;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  bool failed = false;
  std::vector<const void *> ptr;
  std::vector<long> len;
  explicit MockWorld(int nranks) : n(nranks), ptr(nranks, nullptr), len(nranks, 0) {}
  void barrier();          // throws std::runtime_error in every waiting rank once some rank called fail()
  void fail();
};
struct MockCommRank { MockWorld *w; int rank; };
typedef MockCommRank *MPI_Comm;
typedef int MPI_Datatype;
typedef int MPI_Op;
enum { MPI_SUCCESS = 0 };
enum { MPI_DOUBLE = 1, MPI_INT = 2, MPI_BYTE = 3 };
enum { MPI_SUM = 1, MPI_MAX = 2 };
#define MPI_IN_PLACE (reinterpret_cast<void *>(1))

int MPI_Comm_rank(MPI_Comm, int *);
int MPI_Comm_size(MPI_Comm, int *);
int MPI_Barrier(MPI_Comm);
int MPI_Allreduce(const void *send, void *recv, int count, MPI_Datatype, MPI_Op, MPI_Comm);
int MPI_Allgather(const void *send, int scount, MPI_Datatype, void *recv, int rcount, MPI_Datatype, MPI_Comm);
int MPI_Allgatherv(const void *send, int scount, MPI_Datatype, void *recv, const int *rcounts, const int *displs, MPI_Datatype, MPI_Comm);
