// TEST INFRASTRUCTURE ONLY (tests/).  Not part of the product; bench.py never loads it.
//
// A stand-in for librccl that lets the product's RCCL path (csrc/c2ray_comm.inc, c2r_comm_kind == 1: grouped
// ncclAllReduce calls for several communicators, per-slab events on several contexts, the tail sum, one host thread
// per device) run with 2..16 ranks on ONE device of a one-GPU box, where the real RCCL refuses duplicate devices.
// It exports the ten entry points c2ray_comm.inc binds, with the declarations of <rccl/rccl.h>:
//
//   ncclGetVersion            the version code of the header this file was compiled against
//   ncclGetUniqueId           128 bytes that name a clique inside THIS process
//   ncclCommInitAll           n communicators of one clique, any devices (repeats allowed: that is the point)
//   ncclCommInitRank          joins the clique named by the id; returns when all nranks have joined (inside a group:
//                             at ncclGroupEnd), as the real call does
//   ncclAllReduce             ncclFloat64 + ncclSum only.  The k-th call on each communicator of a clique is one
//                             collective.  Every participant records an event on ITS stream (what came before on that
//                             stream is input); when the last one has arrived a kernel on the clique's own stream waits
//                             for all of those events and writes  ((s_0 + s_1) + s_2) + ...  -- RANK ORDER, so a test
//                             can name the association -- into every receive buffer; every participant's stream then
//                             waits for that kernel.  Stream order is honoured on both sides; the HOST blocks in the
//                             call (or in ncclGroupEnd) until the collective has been launched, i.e. until every rank
//                             has issued it -- stricter than RCCL, whose calls return at once, so whatever runs here
//                             without a deadlock does there.
//   ncclGroupStart / End      calls between them are registered first and waited for afterwards: one thread may issue
//                             the same collective for several communicators (c2r_create_multi)
//   ncclCommAbort             marks the clique broken: every rank blocked in a collective, and every later call, gets
//                             ncclRemoteError
//   ncclCommDestroy, ncclGetErrorString
//
// FAKE_RCCL_MULTIPROCESS=1: ranks in SEVERAL PROCESSES on the one device, one communicator per process -- the launch shape of
// torch.distributed.run / MPI (ncclGetUniqueId names a POSIX shared-memory block; ncclCommInitRank maps it).  Collectives are
// then fully host-synchronous: a rank waits for its stream, announces its buffers (hipIpcGetMemHandle), the last to arrive
// maps the others' buffers (hipIpcOpenMemHandle), runs the same rank-ordered sum, waits for it, and lets everybody go.  No
// overlap of anything -- it exercises control flow and data flow between processes, nothing else.
// Otherwise ranks live in one process (threads, or one thread for all).  A rank that never arrives
// makes the others give up after FAKE_RCCL_TIMEOUT_S seconds (default 60) with ncclSystemError -- a test fails, it
// does not hang.  FAKE_RCCL_HANG=1: a rank whose peers are missing (or whose clique was aborted by another rank) does
// NOT get an error: the call returns ncclSuccess and the caller's stream is left waiting behind a host function that
// only ncclCommAbort on that rank's own communicator (or the time-out) releases -- what a real RCCL kernel does when a
// peer has died, and what the product's watchdog (C2R_COMM_TIMEOUT_S) is there for.
//
// fake_rccl_stats(out[4]): cliques made, ncclAllReduce calls, collectives launched, largest clique -- so that a test can
// check that THIS library carried the sums.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unistd.h>
#include <vector>

#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>

namespace {

constexpr int MAXR = 16;

struct Ptrs {
  const double *s[MAXR];
  double *r[MAXR];
};

__global__ void __launch_bounds__(256) k_sum_rank_order(Ptrs p, int n, size_t count) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    double a = p.s[0][i];
    for (int k = 1; k < n; k++) a = a + p.s[k][i];
    for (int k = 0; k < n; k++) p.r[k][i] = a;
  }
}

struct Clique;
struct ShmClique;
struct Comm {
  ShmClique *mp = nullptr; // FAKE_RCCL_MULTIPROCESS: the clique lives in shared memory
  std::string mp_name;
  int mp_n = 0;
  unsigned long long mp_seq = 0;
  std::shared_ptr<Clique> q;
  int rank = 0, device = 0;
  unsigned long long seq = 0; // collectives issued on this communicator
  bool gone = false;
  // FAKE_RCCL_HANG: what the caller's stream was left waiting behind
  std::mutex hm;
  std::condition_variable hcv;
  bool hang_released = false;
  int hangs = 0;
  hipStream_t hang_stream = nullptr;
};

ncclResult_t mp_retire(Comm *c, bool abort); // (multi-process mode, below)

struct Op {
  int arrived = 0, released = 0;
  bool launched = false;
  size_t count = 0;
  Ptrs p{};
  hipStream_t stream[MAXR] = {};
  hipEvent_t ev_in[MAXR] = {};
  hipEvent_t ev_out = nullptr;
};

struct Clique {
  int n = 0, joined = 0, alive = 0, device = -1;
  bool broken = false;
  std::mutex m;
  std::condition_variable cv;
  std::map<unsigned long long, Op> ops;
  hipStream_t work = nullptr;
  std::vector<hipEvent_t> events; // destroyed with the clique
  std::vector<Comm *> comms;
  std::string id;
};

std::mutex g_m;
std::map<std::string, std::shared_ptr<Clique>> g_by_id;
unsigned long long g_ids = 0;
std::atomic<long long> g_stats[4];

struct Pending {
  int kind; // 0: join of a clique, 1: all-reduce
  Comm *c;
  unsigned long long seq;
};
thread_local int t_depth = 0;
thread_local std::vector<Pending> t_pending;
thread_local ncclResult_t t_group_error = ncclSuccess;

double timeout_s() {
  const char *e = getenv("FAKE_RCCL_TIMEOUT_S");
  return e && atof(e) > 0 ? atof(e) : 60.0;
}
bool hang_mode() {
  const char *e = getenv("FAKE_RCCL_HANG");
  return e && atoi(e) > 0;
}

struct DeviceGuard { // the real library leaves the caller's current device alone
  int prev = -1;
  explicit DeviceGuard(int dev) {
    (void)hipGetDevice(&prev);
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    int now = -1;
    (void)hipGetDevice(&now);
    if (prev >= 0 && now != prev) (void)hipSetDevice(prev);
  }
};

// all ranks have arrived (clique lock held): the sum, on the clique's own stream
ncclResult_t launch(Clique &q, Op &op) {
  if (!q.work && hipStreamCreateWithFlags(&q.work, hipStreamNonBlocking) != hipSuccess) return ncclUnhandledCudaError;
  for (int r = 0; r < q.n; r++)
    if (hipStreamWaitEvent(q.work, op.ev_in[r], 0) != hipSuccess) return ncclUnhandledCudaError;
  const int nblk = (int)std::min<size_t>(8192, (op.count + 255) / 256);
  if (nblk > 0) hipLaunchKernelGGL(k_sum_rank_order, dim3(nblk), dim3(256), 0, q.work, op.p, q.n, op.count);
  if (hipGetLastError() != hipSuccess) return ncclUnhandledCudaError;
  if (hipEventCreateWithFlags(&op.ev_out, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
  q.events.push_back(op.ev_out);
  if (hipEventRecord(op.ev_out, q.work) != hipSuccess) return ncclUnhandledCudaError;
  op.launched = true;
  g_stats[2]++;
  q.cv.notify_all();
  return ncclSuccess;
}

// FAKE_RCCL_HANG: leave `stream` waiting until this communicator is aborted (or the time-out passes)
void hang_host_fn(void *arg) {
  Comm *c = static_cast<Comm *>(arg);
  std::unique_lock<std::mutex> lk(c->hm);
  c->hcv.wait_for(lk, std::chrono::duration<double>(timeout_s()), [c] { return c->hang_released; });
}
ncclResult_t leave_hanging(Comm *c, hipStream_t stream) {
  DeviceGuard g(c->device);
  if (!c->hang_stream && hipStreamCreateWithFlags(&c->hang_stream, hipStreamNonBlocking) != hipSuccess) return ncclUnhandledCudaError;
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
  if (hipLaunchHostFunc(c->hang_stream, hang_host_fn, c) != hipSuccess) return ncclUnhandledCudaError;
  if (hipEventRecord(e, c->hang_stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipStreamWaitEvent(stream, e, 0) != hipSuccess) return ncclUnhandledCudaError;
  c->hangs++;
  return ncclSuccess; // the event is leaked: a failure path of a test
}

ncclResult_t wait_joined(Comm *c) {
  Clique &q = *c->q;
  std::unique_lock<std::mutex> lk(q.m);
  const bool ok = q.cv.wait_for(lk, std::chrono::duration<double>(timeout_s()), [&] { return q.joined == q.n || q.broken; });
  if (q.broken) return ncclRemoteError;
  return ok ? ncclSuccess : ncclSystemError;
}

ncclResult_t wait_launched(Comm *c, unsigned long long seq) {
  Clique &q = *c->q;
  hipStream_t mine = nullptr;
  {
    std::unique_lock<std::mutex> lk(q.m);
    auto it = q.ops.find(seq);
    if (it == q.ops.end()) return ncclInternalError;
    Op &op = it->second;
    mine = op.stream[c->rank];
    const double patience = hang_mode() ? std::min(timeout_s(), 3.0) : timeout_s();
    const bool ok = q.cv.wait_for(lk, std::chrono::duration<double>(patience), [&] { return op.launched || q.broken; });
    if (op.launched) {
      DeviceGuard g(c->device);
      const hipError_t e = hipStreamWaitEvent(mine, op.ev_out, 0);
      if (++op.released == q.n) q.ops.erase(it);
      return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
    }
    if (!hang_mode()) return q.broken ? ncclRemoteError : (ok ? ncclInternalError : ncclSystemError);
  }
  return leave_hanging(c, mine); // peers missing: behave like a kernel that waits for them
}

ncclResult_t settle(const std::vector<Pending> &list) {
  ncclResult_t rc = ncclSuccess;
  for (const Pending &p : list) {
    const ncclResult_t r = p.kind == 0 ? wait_joined(p.c) : wait_launched(p.c, p.seq);
    if (rc == ncclSuccess) rc = r;
  }
  return rc;
}

void free_clique(Clique &q) {
  if (q.device >= 0) (void)hipSetDevice(q.device);
  if (q.work) {
    (void)hipStreamSynchronize(q.work);
    (void)hipStreamDestroy(q.work);
    q.work = nullptr;
  }
  for (hipEvent_t e : q.events) (void)hipEventDestroy(e);
  q.events.clear();
  q.ops.clear();
}

ncclResult_t retire(Comm *c, bool abort) {
  if (!c || c->gone) return ncclInvalidArgument;
  if (c->mp) return mp_retire(c, abort);
  std::shared_ptr<Clique> q = c->q;
  {
    std::lock_guard<std::mutex> lk(c->hm);
    c->hang_released = true;
  }
  c->hcv.notify_all();
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (c->hang_stream) {
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->hang_stream);
    (void)hipStreamDestroy(c->hang_stream);
    c->hang_stream = nullptr;
  }
  bool last = false;
  {
    std::lock_guard<std::mutex> lk(q->m);
    if (abort && !q->broken) {
      q->broken = true;
      q->cv.notify_all();
    }
    c->gone = true;
    last = --q->alive == 0;
    if (last) free_clique(*q);
  }
  if (last) {
    std::lock_guard<std::mutex> lk(g_m);
    g_by_id.erase(q->id);
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  return ncclSuccess; // the Comm itself is leaked on purpose: a late call on it must find `gone`, not freed memory
}

// ---------------------------------------------------------------------------------------------------------------------
// ranks in several processes (FAKE_RCCL_MULTIPROCESS=1)
struct ShmRank {
  hipIpcMemHandle_t send_h, recv_h;
  size_t send_off, recv_off;
};
struct ShmClique {
  pthread_mutex_t m;
  pthread_cond_t cv;
  int n, joined, gone, broken;
  unsigned long long seq; // the collective in progress
  int arrived, left, done, status;
  size_t count;
  ShmRank r[MAXR];
};

bool multiprocess_mode() {
  const char *e = getenv("FAKE_RCCL_MULTIPROCESS");
  return e && atoi(e) > 0;
}

// cond wait with the stand-in's time-out; false when time is up
bool mp_wait(ShmClique *q, double seconds) {
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  const long long ns = (long long)(seconds * 1e9) + ts.tv_nsec;
  ts.tv_sec += ns / 1000000000LL;
  ts.tv_nsec = ns % 1000000000LL;
  return pthread_cond_timedwait(&q->cv, &q->m, &ts) == 0;
}

ncclResult_t mp_export(const void *p, hipIpcMemHandle_t *h, size_t *off) {
  // (asked for at every collective: an address can belong to another allocation the next time it is seen)
  void *base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, const_cast<void *>(p)) != hipSuccess) return ncclUnhandledCudaError;
  if (hipIpcGetMemHandle(h, base) != hipSuccess) return ncclUnhandledCudaError;
  *off = (size_t)(static_cast<const char *>(p) - static_cast<const char *>(base));
  return ncclSuccess;
}

void *mp_import(const hipIpcMemHandle_t &h, size_t off) {
  static std::mutex m;
  static std::map<std::string, void *> cache; // a peer's allocation, mapped once
  const std::string key(reinterpret_cast<const char *>(&h), sizeof h);
  std::lock_guard<std::mutex> lk(m);
  auto it = cache.find(key);
  if (it == cache.end()) {
    void *p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return nullptr;
    it = cache.emplace(key, p).first;
  }
  return static_cast<char *>(it->second) + off;
}

ncclResult_t mp_allreduce(Comm *c, const void *sendbuff, void *recvbuff, size_t count, hipStream_t stream) {
  ShmClique *q = c->mp;
  DeviceGuard g(c->device);
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError; // my input is complete
  ShmRank mine;
  ncclResult_t rc = mp_export(sendbuff, &mine.send_h, &mine.send_off);
  if (rc == ncclSuccess) rc = mp_export(recvbuff, &mine.recv_h, &mine.recv_off);
  if (rc != ncclSuccess) return rc;
  const unsigned long long seq = c->mp_seq++;
  pthread_mutex_lock(&q->m);
  g_stats[1]++;
  // the slot of the collective before this one must have been left by everybody
  while (!q->broken && q->seq != seq)
    if (!mp_wait(q, timeout_s())) { pthread_mutex_unlock(&q->m); return ncclSystemError; }
  if (q->broken) { pthread_mutex_unlock(&q->m); return ncclRemoteError; }
  if (q->arrived == 0) { q->count = count; q->done = 0; q->status = 0; }
  if (q->count != count) { q->broken = 1; pthread_cond_broadcast(&q->cv); pthread_mutex_unlock(&q->m); return ncclInvalidArgument; }
  q->r[c->rank] = mine;
  if (++q->arrived == q->n) {
    // everybody's input is complete and announced: the sum, here, now
    Ptrs p{};
    bool ok = true;
    for (int r = 0; r < q->n && ok; r++) {
      if (r == c->rank) {
        p.s[r] = static_cast<const double *>(sendbuff);
        p.r[r] = static_cast<double *>(recvbuff);
      } else {
        p.s[r] = static_cast<const double *>(mp_import(q->r[r].send_h, q->r[r].send_off));
        p.r[r] = static_cast<double *>(mp_import(q->r[r].recv_h, q->r[r].recv_off));
        ok = p.s[r] && p.r[r];
      }
    }
    if (ok && count > 0) {
      const int nblk = (int)std::min<size_t>(8192, (count + 255) / 256);
      hipLaunchKernelGGL(k_sum_rank_order, dim3(nblk), dim3(256), 0, stream, p, q->n, count);
      ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
    }
    q->status = ok ? 0 : 1;
    q->done = 1;
    g_stats[2]++;
    pthread_cond_broadcast(&q->cv);
  } else {
    while (!q->done && !q->broken)
      if (!mp_wait(q, timeout_s())) { q->broken = 1; pthread_cond_broadcast(&q->cv); pthread_mutex_unlock(&q->m); return ncclSystemError; }
  }
  const bool bad = q->broken || q->status != 0;
  if (++q->left == q->n) { // the last one out frees the slot for the next collective
    q->arrived = q->left = 0;
    q->seq++;
    pthread_cond_broadcast(&q->cv);
  }
  pthread_mutex_unlock(&q->m);
  return bad ? (q->broken ? ncclRemoteError : ncclUnhandledCudaError) : ncclSuccess;
}

ncclResult_t mp_join(Comm *c, int nranks, const ncclUniqueId &id, int rank) {
  const char *name = std::strchr(id.internal, '/');
  if (!name) return ncclInvalidArgument;
  const int fd = shm_open(name, O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  void *p = mmap(nullptr, sizeof(ShmClique), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  ShmClique *q = static_cast<ShmClique *>(p);
  c->mp = q;
  c->mp_name = name;
  c->mp_n = nranks;
  c->rank = rank;
  (void)hipGetDevice(&c->device);
  pthread_mutex_lock(&q->m);
  if (q->n == 0) q->n = nranks;
  ncclResult_t rc = q->n == nranks ? ncclSuccess : ncclInvalidUsage;
  if (rc == ncclSuccess) {
    q->joined++;
    pthread_cond_broadcast(&q->cv);
    while (q->joined < q->n && !q->broken)
      if (!mp_wait(q, timeout_s())) { rc = ncclSystemError; break; }
    if (q->broken) rc = ncclRemoteError;
  }
  pthread_mutex_unlock(&q->m);
  {
    std::lock_guard<std::mutex> lk(g_m);
    g_stats[0]++;
    if (g_stats[3] < nranks) g_stats[3] = nranks;
  }
  return rc;
}

ncclResult_t mp_retire(Comm *c, bool abort) {
  ShmClique *q = c->mp;
  pthread_mutex_lock(&q->m);
  if (abort) q->broken = 1;
  const bool last = ++q->gone == q->n;
  pthread_cond_broadcast(&q->cv);
  pthread_mutex_unlock(&q->m);
  if (last) shm_unlink(c->mp_name.c_str());
  munmap(q, sizeof(ShmClique));
  c->mp = nullptr;
  c->gone = true;
  return ncclSuccess;
}

} // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) {
  if (!version) return ncclInvalidArgument;
  *version = NCCL_VERSION_CODE;
  return ncclSuccess;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lk(g_m);
  std::memset(id->internal, 0, sizeof id->internal);
  if (multiprocess_mode()) {
    // the clique is a block of shared memory that this call makes and the last rank to leave removes
    snprintf(id->internal, sizeof id->internal, "FAKE-RCCL-MP /fake_rccl_%d_%llu", (int)getpid(), ++g_ids);
    const char *name = std::strchr(id->internal, '/');
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(ShmClique)) != 0) return ncclSystemError;
    void *p = mmap(nullptr, sizeof(ShmClique), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    ShmClique *q = static_cast<ShmClique *>(p);
    std::memset(q, 0, sizeof *q);
    pthread_mutexattr_t ma;
    pthread_mutexattr_init(&ma);
    pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
    pthread_mutex_init(&q->m, &ma);
    pthread_condattr_t ca;
    pthread_condattr_init(&ca);
    pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
    pthread_cond_init(&q->cv, &ca);
    munmap(q, sizeof(ShmClique));
    return ncclSuccess;
  }
  snprintf(id->internal, sizeof id->internal, "FAKE-RCCL clique %llu of process %d", ++g_ids, (int)getpid());
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
  if (!comms || ndev < 1 || ndev > MAXR) return ncclInvalidArgument;
  auto q = std::make_shared<Clique>();
  q->n = q->joined = q->alive = ndev;
  q->comms.resize((size_t)ndev);
  {
    std::lock_guard<std::mutex> lk(g_m);
    q->id = "local clique " + std::to_string(++g_ids);
    g_by_id[q->id] = q;
    g_stats[0]++;
    if (g_stats[3] < ndev) g_stats[3] = ndev;
  }
  for (int i = 0; i < ndev; i++) {
    Comm *c = new Comm;
    c->q = q;
    c->rank = i;
    c->device = devlist ? devlist[i] : i;
    q->comms[(size_t)i] = c;
    comms[i] = reinterpret_cast<ncclComm_t>(c);
  }
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  if (std::strncmp(id.internal, "FAKE-RCCL", 9) != 0) return ncclInvalidArgument; // an id of another library
  if (std::strncmp(id.internal, "FAKE-RCCL-MP", 12) == 0) {
    Comm *c = new Comm;
    *comm = reinterpret_cast<ncclComm_t>(c);
    return mp_join(c, nranks, id, rank); // (inside a group too: one communicator per process, nothing to defer)
  }
  std::shared_ptr<Clique> q;
  {
    std::lock_guard<std::mutex> lk(g_m);
    const std::string key(id.internal, sizeof id.internal);
    auto it = g_by_id.find(key);
    if (it == g_by_id.end()) {
      q = std::make_shared<Clique>();
      q->n = nranks;
      q->id = key;
      q->comms.assign((size_t)nranks, nullptr);
      g_by_id[key] = q;
      g_stats[0]++;
      if (g_stats[3] < nranks) g_stats[3] = nranks;
    } else {
      q = it->second;
    }
  }
  Comm *c = new Comm;
  {
    std::lock_guard<std::mutex> lk(q->m);
    if (q->n != nranks || q->comms[(size_t)rank] || q->broken) {
      delete c;
      return ncclInvalidUsage;
    }
    c->q = q;
    c->rank = rank;
    (void)hipGetDevice(&c->device);
    q->comms[(size_t)rank] = c;
    q->joined++;
    q->alive++;
    q->cv.notify_all();
  }
  *comm = reinterpret_cast<ncclComm_t>(c);
  if (t_depth > 0) {
    t_pending.push_back(Pending{0, c, 0});
    return ncclSuccess;
  }
  return wait_joined(c);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c || c->gone || !sendbuff || !recvbuff) return ncclInvalidArgument;
  if (datatype != ncclFloat64 || op != ncclSum) return ncclInvalidArgument; // all the product ever asks for
  if (c->mp) {
    const ncclResult_t r = mp_allreduce(c, sendbuff, recvbuff, count, stream);
    if (r != ncclSuccess && t_depth > 0 && t_group_error == ncclSuccess) t_group_error = r;
    return r;
  }
  Clique &q = *c->q;
  ncclResult_t rc = ncclSuccess;
  unsigned long long seq = 0;
  bool registered = false;
  {
    std::lock_guard<std::mutex> lk(q.m);
    g_stats[1]++;
    seq = c->seq++;
    if (q.broken) {
      rc = ncclRemoteError;
    } else if (q.joined != q.n) {
      rc = ncclInvalidUsage;
    } else {
      if (q.device < 0) q.device = c->device;
      if (q.device != c->device) {
        rc = ncclInvalidUsage; // the stand-in sums on ONE device
      } else {
        Op &o = q.ops[seq];
        if (o.arrived == 0) o.count = count;
        if (o.count != count) {
          q.broken = true;
          q.cv.notify_all();
          rc = ncclInvalidArgument;
        } else {
          DeviceGuard g(c->device);
          o.p.s[c->rank] = static_cast<const double *>(sendbuff);
          o.p.r[c->rank] = static_cast<double *>(recvbuff);
          o.stream[c->rank] = stream;
          if (hipEventCreateWithFlags(&o.ev_in[c->rank], hipEventDisableTiming) != hipSuccess) rc = ncclUnhandledCudaError;
          else {
            q.events.push_back(o.ev_in[c->rank]);
            if (hipEventRecord(o.ev_in[c->rank], stream) != hipSuccess) rc = ncclUnhandledCudaError;
          }
          registered = true;
          if (rc == ncclSuccess && ++o.arrived == q.n) rc = launch(q, o);
        }
      }
    }
  }
  if (rc == ncclRemoteError && hang_mode() && !registered) return leave_hanging(c, stream);
  if (rc != ncclSuccess) {
    if (t_depth > 0 && t_group_error == ncclSuccess) t_group_error = rc;
    return rc;
  }
  if (t_depth > 0) {
    t_pending.push_back(Pending{1, c, seq});
    return ncclSuccess;
  }
  return wait_launched(c, seq);
}

ncclResult_t ncclGroupStart(void) {
  if (t_depth++ == 0) t_group_error = ncclSuccess;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  std::vector<Pending> list;
  list.swap(t_pending);
  const ncclResult_t rc = settle(list);
  return t_group_error != ncclSuccess ? t_group_error : rc;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { return retire(reinterpret_cast<Comm *>(comm), false); }
ncclResult_t ncclCommAbort(ncclComm_t comm) { return retire(reinterpret_cast<Comm *>(comm), true); }

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake RCCL)";
    case ncclSystemError: return "a rank did not arrive in time (fake RCCL, FAKE_RCCL_TIMEOUT_S)";
    case ncclInternalError: return "internal error (fake RCCL)";
    case ncclInvalidArgument: return "invalid argument (fake RCCL)";
    case ncclInvalidUsage: return "invalid usage (fake RCCL)";
    case ncclRemoteError: return "a peer aborted its communicator (fake RCCL)";
    default: return "error (fake RCCL)";
  }
}

// test hook, not part of RCCL: {cliques made, ncclAllReduce calls, collectives launched, largest clique}
void fake_rccl_stats(long long out[4]) {
  for (int i = 0; i < 4; i++) out[i] = g_stats[i].load();
}

} // extern "C"
