// comm.cpp -- the cross-GPU exchange of the path-integral loop on RCCL (part of libmpmc_energy.so; C ABI in include/mpmc_energy.h).
//
// The reference combines the P beads with 4 x MPI_Allgather of one double per rank and an ordered sum s = 0..P-1
// (src/SimulationControl.PathIntegral.cpp:763-766, :786-801).  Here a bead is a device context; the exchange is ONE ncclAllGather of
// `stride` fp64 per bead over xGMI, followed by the same ordered sum on the host -- bit-identical on every rank, which an
// ncclAllReduce (ring order) is not.  Two ways to own GPUs, one code path:
//   * one process per GPU  : mpmc_comm_unique_id (rank 0) -> the host program hands the 128 bytes to the other ranks (MPI, a file, the
//                            torch.distributed store ...) -> mpmc_comm_init_rank everywhere;
//   * one process, G GPUs  : mpmc_comm_init_all (ncclCommInitAll), bead b on device b mod G; mpmc_pi_allreduce drives
//                            evaluation and combine with one host thread per device.
// RCCL is opened with dlopen at first use (see rccl() for which copy): the energy path itself has no link-time dependency on it, and a
// host program that already carries an RCCL (PyTorch does) shares that copy.  Messages are 32 B .. a few KiB: latency-bound, link bandwidth is irrelevant.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "context.h"

using namespace mpmc;

namespace {

struct RcclApi {
	void *handle = nullptr;
	std::string error, path, how, described;
	decltype(&ncclGetVersion) GetVersion = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi g_rccl;
std::string rccl_error_text() { return g_rccl.error.empty() ? std::string("librccl.so not found") : g_rccl.error; }

// directory of the HIP runtime this library is bound to (the copy the dynamic linker resolved libamdhip64.so.7 to in THIS process)
static std::string dir_of_bound_hip_runtime() {
	Dl_info info;
	if (!dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) || !info.dli_fname) return "";
	std::string f = info.dli_fname;
	size_t cut = f.rfind('/');
	return cut == std::string::npos ? std::string() : f.substr(0, cut);
}

// Which RCCL (one per process, and the one that belongs to the HIP runtime in use):
//   1. $MPMC_RCCL_LIB, if set: the host program's explicit choice;
//   2. a librccl.so.1 that is ALREADY mapped (RTLD_NOLOAD): a host that carries an RCCL -- PyTorch does -- shares its copy, so the
//      process keeps one RCCL and one librocm_smi64 underneath it;
//   3. the librccl next to the libamdhip64 this library is bound to: same ROCm release as the runtime whose streams and buffers it gets;
//   4. the loader's search path.
// Always RTLD_LOCAL.  Round 4 opened /opt/rocm's copy RTLD_GLOBAL: its librocm_smi64.so.1 then sat in the global scope, the
// librocm_smi64.so.7 of a PyTorch imported LATER bound its exported statics (amd::smi::Device::devInfoTypesStrings) to the first copy, and
// both ran the destructor at exit -- "double free or corruption" (tests/test_rccl_loading.py replays that order in a child process).
RcclApi *rccl() {
	RcclApi &api = g_rccl;
	static std::once_flag once;
	std::call_once(once, [&api] {
		struct Cand {
			std::string path;
			int flags;
			const char *how;
		};
		std::vector<Cand> cand;
		if (const char *e = std::getenv("MPMC_RCCL_LIB"))
			if (*e) cand.push_back({e, RTLD_NOW | RTLD_LOCAL, "MPMC_RCCL_LIB"});
		for (const char *nm : {"librccl.so.1", "librccl.so"}) cand.push_back({nm, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD, "already mapped by the host program"});
		const std::string hipdir = dir_of_bound_hip_runtime();
		if (!hipdir.empty())
			for (const char *nm : {"/librccl.so.1", "/librccl.so"}) cand.push_back({hipdir + nm, RTLD_NOW | RTLD_LOCAL, "next to the bound libamdhip64"});
		for (const char *nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) cand.push_back({nm, RTLD_NOW | RTLD_LOCAL, "loader search path"});
		for (const Cand &c : cand) {
			api.handle = dlopen(c.path.c_str(), c.flags);
			if (api.handle) {
				api.path = c.path;
				api.how = c.how;
				break;
			}
			if (!(c.flags & RTLD_NOLOAD)) api.error = dlerror();
		}
		if (!api.handle) return;
#define MPMC_SYM(name)                                                             \
	api.name = reinterpret_cast<decltype(api.name)>(dlsym(api.handle, "nccl" #name)); \
	if (!api.name) {                                                                \
		api.error = "RCCL symbol nccl" #name " not found in " + api.path;             \
		api.handle = nullptr;                                                       \
		return;                                                                     \
	}
		MPMC_SYM(GetVersion)
		MPMC_SYM(GetUniqueId)
		MPMC_SYM(CommInitRank)
		MPMC_SYM(CommInitAll)
		MPMC_SYM(CommDestroy)
		MPMC_SYM(AllGather)
		MPMC_SYM(GroupStart)
		MPMC_SYM(GroupEnd)
		MPMC_SYM(GetErrorString)
#undef MPMC_SYM
		Dl_info info; // the file the symbols really come from (a bare soname says nothing)
		if (dladdr(reinterpret_cast<void *>(api.GetVersion), &info) && info.dli_fname) api.path = info.dli_fname;
		api.described = api.path + " (" + api.how + ")";
	});
	return api.handle ? &api : nullptr;
}

thread_local std::string g_comm_error;
int comm_fail(int code, const std::string &msg) {
	g_comm_error = msg;
	g_create_error = msg; // mpmc_last_error(NULL)
	return code;
}

} // namespace

// one member per device this process drives: 1 with one process per GPU, G after mpmc_comm_init_all
struct CommMember {
	int device = 0;
	ncclComm_t comm = nullptr;
	hipStream_t stream = nullptr;
	double *d_send = nullptr, *d_recv = nullptr;
	double *h_send = nullptr, *h_recv = nullptr; // pinned
	size_t cap = 0;                              // doubles per rank the buffers hold
};
struct mpmc_comm {
	int n_ranks = 1;  // size of the communicator (processes x devices per process)
	int rank0 = 0;    // communicator rank of members[0]; members[g] is rank0 + g
	std::vector<CommMember> members;
	std::string err;
	bool host_only = false; // members are "virtual devices" of one GPU (test hook, see group_beads): the gather is a host copy, no RCCL underneath
};

#define RCCL_TRY(cm, api, call)                                                                      \
	do {                                                                                             \
		ncclResult_t _r = (call);                                                                    \
		if (_r != ncclSuccess) {                                                                     \
			(cm)->err = std::string(#call) + ": " + (api)->GetErrorString(_r);                       \
			return comm_fail(MPMC_ERR_COMM, (cm)->err);                                              \
		}                                                                                            \
	} while (0)
#define HIPC_TRY(cm, call)                                                                           \
	do {                                                                                             \
		hipError_t _e = (call);                                                                      \
		if (_e != hipSuccess) {                                                                      \
			(cm)->err = std::string(#call) + ": " + hipGetErrorString(_e);                           \
			return comm_fail(MPMC_ERR_HIP, (cm)->err);                                               \
		}                                                                                            \
	} while (0)

static int member_reserve(mpmc_comm *cm, CommMember &m, size_t per_rank) {
	if (per_rank <= m.cap) return MPMC_OK;
	HIPC_TRY(cm, hipSetDevice(m.device));
	if (m.d_send) (void)hipFree(m.d_send);
	if (m.d_recv) (void)hipFree(m.d_recv);
	if (m.h_send) (void)pinned_free(m.h_send);
	if (m.h_recv) (void)pinned_free(m.h_recv);
	m.d_send = m.d_recv = m.h_send = m.h_recv = nullptr;
	m.cap = 0;
	const size_t cap = std::max<size_t>(per_rank, 64);
	HIPC_TRY(cm, hipMalloc((void **)&m.d_send, cap * sizeof(double)));
	HIPC_TRY(cm, hipMalloc((void **)&m.d_recv, cap * (size_t)cm->n_ranks * sizeof(double)));
	HIPC_TRY(cm, pinned_alloc(&m.h_send, cap * sizeof(double)));
	HIPC_TRY(cm, pinned_alloc(&m.h_recv, cap * (size_t)cm->n_ranks * sizeof(double)));
	m.cap = cap;
	return MPMC_OK;
}

extern "C" const char *mpmc_comm_last_error(const mpmc_comm *cm) { return cm ? cm->err.c_str() : g_comm_error.c_str(); }

extern "C" int mpmc_rccl_version(int *version) {
	if (!version) return MPMC_ERR_ARG;
	*version = 0;
	RcclApi *api = rccl();
	if (!api) return comm_fail(MPMC_ERR_COMM, "RCCL could not be loaded: " + rccl_error_text());
	return api->GetVersion(version) == ncclSuccess ? MPMC_OK : comm_fail(MPMC_ERR_COMM, "ncclGetVersion failed");
}

extern "C" const char *mpmc_rccl_library_path(void) {
	RcclApi *api = rccl();
	return api ? api->described.c_str() : "";
}

extern "C" int mpmc_comm_unique_id(char id[MPMC_COMM_ID_BYTES]) {
	static_assert(MPMC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "mpmc_energy.h carries RCCL's unique-id size");
	if (!id) return MPMC_ERR_ARG;
	RcclApi *api = rccl();
	if (!api) return comm_fail(MPMC_ERR_COMM, "RCCL could not be loaded: " + rccl_error_text());
	ncclUniqueId u;
	ncclResult_t r = api->GetUniqueId(&u);
	if (r != ncclSuccess) return comm_fail(MPMC_ERR_COMM, std::string("ncclGetUniqueId: ") + api->GetErrorString(r));
	std::memcpy(id, u.internal, MPMC_COMM_ID_BYTES);
	return MPMC_OK;
}

static int finish_members(mpmc_comm *cm) {
	for (CommMember &m : cm->members) {
		HIPC_TRY(cm, hipSetDevice(m.device));
		HIPC_TRY(cm, hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));
		int rc = member_reserve(cm, m, 64);
		if (rc != MPMC_OK) return rc;
	}
	return MPMC_OK;
}

extern "C" int mpmc_comm_init_rank(mpmc_comm **out, int n_ranks, int rank, const char id[MPMC_COMM_ID_BYTES], int device) {
	if (!out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return comm_fail(MPMC_ERR_ARG, "mpmc_comm_init_rank: bad argument");
	*out = nullptr;
	RcclApi *api = rccl();
	if (!api) return comm_fail(MPMC_ERR_COMM, "RCCL could not be loaded: " + rccl_error_text());
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
		return comm_fail(MPMC_ERR_NO_DEVICE, "mpmc_comm_init_rank: no such HIP device");
	mpmc_comm *cm = new mpmc_comm();
	cm->n_ranks = n_ranks;
	cm->rank0 = rank;
	cm->members.resize(1);
	cm->members[0].device = device;
	ncclUniqueId u;
	std::memcpy(u.internal, id, MPMC_COMM_ID_BYTES);
	int rc = MPMC_OK;
	if (hipSetDevice(device) != hipSuccess) rc = comm_fail(MPMC_ERR_NO_DEVICE, "mpmc_comm_init_rank: hipSetDevice failed");
	if (rc == MPMC_OK) {
		ncclResult_t r = api->CommInitRank(&cm->members[0].comm, n_ranks, u, rank);
		if (r != ncclSuccess) rc = comm_fail(MPMC_ERR_COMM, std::string("ncclCommInitRank: ") + api->GetErrorString(r));
	}
	if (rc == MPMC_OK) rc = finish_members(cm);
	if (rc != MPMC_OK) {
		mpmc_comm_destroy(cm);
		return rc;
	}
	*out = cm;
	return MPMC_OK;
}

extern "C" int mpmc_comm_init_all(mpmc_comm **out, int n_devices, const int *devices) {
	if (!out || n_devices < 1) return comm_fail(MPMC_ERR_ARG, "mpmc_comm_init_all: bad argument");
	*out = nullptr;
	RcclApi *api = rccl();
	if (!api) return comm_fail(MPMC_ERR_COMM, "RCCL could not be loaded: " + rccl_error_text());
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return comm_fail(MPMC_ERR_NO_DEVICE, "mpmc_comm_init_all: no HIP device");
	std::vector<int> devs(n_devices);
	for (int g = 0; g < n_devices; g++) {
		devs[g] = devices ? devices[g] : g;
		if (devs[g] < 0 || devs[g] >= ndev) return comm_fail(MPMC_ERR_ARG, "mpmc_comm_init_all: device index out of range");
		for (int h = 0; h < g; h++)
			if (devs[h] == devs[g]) return comm_fail(MPMC_ERR_ARG, "mpmc_comm_init_all: a device may appear once (RCCL: one rank per GPU)");
	}
	mpmc_comm *cm = new mpmc_comm();
	cm->n_ranks = n_devices;
	cm->rank0 = 0;
	cm->members.resize(n_devices);
	std::vector<ncclComm_t> comms(n_devices, nullptr);
	ncclResult_t r = api->CommInitAll(comms.data(), n_devices, devs.data());
	int rc = MPMC_OK;
	if (r != ncclSuccess) rc = comm_fail(MPMC_ERR_COMM, std::string("ncclCommInitAll: ") + api->GetErrorString(r));
	for (int g = 0; g < n_devices; g++) {
		cm->members[g].device = devs[g];
		cm->members[g].comm = comms[g];
	}
	if (rc == MPMC_OK) rc = finish_members(cm);
	if (rc != MPMC_OK) {
		mpmc_comm_destroy(cm);
		return rc;
	}
	*out = cm;
	return MPMC_OK;
}

extern "C" int mpmc_comm_destroy(mpmc_comm *cm) {
	if (!cm) return MPMC_ERR_ARG;
	RcclApi *api = rccl();
	for (CommMember &m : cm->members) {
		(void)hipSetDevice(m.device);
		if (m.stream) (void)hipStreamSynchronize(m.stream);
		if (m.comm && api) (void)api->CommDestroy(m.comm);
		if (m.d_send) (void)hipFree(m.d_send);
		if (m.d_recv) (void)hipFree(m.d_recv);
		if (m.h_send) (void)pinned_free(m.h_send);
		if (m.h_recv) (void)pinned_free(m.h_recv);
		if (m.stream) (void)hipStreamDestroy(m.stream);
	}
	delete cm;
	return MPMC_OK;
}

extern "C" int mpmc_comm_info(const mpmc_comm *cm, int *n_ranks, int *rank, int *n_local_devices) {
	if (!cm) return MPMC_ERR_ARG;
	if (n_ranks) *n_ranks = cm->n_ranks;
	if (rank) *rank = cm->rank0;
	if (n_local_devices) *n_local_devices = (int)cm->members.size();
	return MPMC_OK;
}

// every member contributes `count` doubles (local[g] for member g); all[g] receives n_ranks x count doubles in rank order.
// One ncclAllGather per member inside a group (RCCL's rule for several devices in one thread).
static int allgather_members(mpmc_comm *cm, const std::vector<const double *> &local, size_t count, const std::vector<double *> &all) {
	const int G = (int)cm->members.size();
	if (cm->host_only) { // every member's block to every member that asked, in member order (what ncclAllGather delivers)
		for (int g = 0; g < G; g++)
			if (all[g])
				for (int h = 0; h < G; h++) std::memcpy(all[g] + (size_t)h * count, local[h], count * sizeof(double));
		return MPMC_OK;
	}
	RcclApi *api = rccl();
	if (!api) return comm_fail(MPMC_ERR_COMM, "RCCL could not be loaded");
	for (int g = 0; g < G; g++) {
		CommMember &m = cm->members[g];
		int rc = member_reserve(cm, m, count);
		if (rc != MPMC_OK) return rc;
		HIPC_TRY(cm, hipSetDevice(m.device));
		std::memcpy(m.h_send, local[g], count * sizeof(double));
		HIPC_TRY(cm, hipMemcpyAsync(m.d_send, m.h_send, count * sizeof(double), hipMemcpyHostToDevice, m.stream));
	}
	RCCL_TRY(cm, api, api->GroupStart());
	for (int g = 0; g < G; g++) {
		CommMember &m = cm->members[g];
		HIPC_TRY(cm, hipSetDevice(m.device));
		RCCL_TRY(cm, api, api->AllGather(m.d_send, m.d_recv, count, ncclDouble, m.comm, m.stream));
	}
	RCCL_TRY(cm, api, api->GroupEnd());
	for (int g = 0; g < G; g++) {
		CommMember &m = cm->members[g];
		if (!all[g]) continue;
		HIPC_TRY(cm, hipSetDevice(m.device));
		HIPC_TRY(cm, hipMemcpyAsync(m.h_recv, m.d_recv, count * (size_t)cm->n_ranks * sizeof(double), hipMemcpyDeviceToHost, m.stream));
	}
	for (int g = 0; g < G; g++) {
		CommMember &m = cm->members[g];
		HIPC_TRY(cm, hipSetDevice(m.device));
		HIPC_TRY(cm, hipStreamSynchronize(m.stream));
		if (all[g]) std::memcpy(all[g], m.h_recv, count * (size_t)cm->n_ranks * sizeof(double));
	}
	return MPMC_OK;
}

extern "C" int mpmc_comm_allgather_f64(mpmc_comm *cm, const double *local, int64_t count, double *all) {
	if (!cm || !local || !all || count <= 0) return MPMC_ERR_ARG;
	if (cm->members.size() != 1) return comm_fail(MPMC_ERR_ARG, "mpmc_comm_allgather_f64: communicator drives several devices (use mpmc_pi_allreduce)");
	return allgather_members(cm, {local}, (size_t)count, {all});
}

// bead s lives on rank s % n_ranks, local slot s / n_ranks (round-robin): rank-major gather -> bead order
extern "C" int mpmc_pi_gather_beads(mpmc_comm *cm, const double *local, int n_local, int stride, double *all) {
	if (!cm || !local || !all || n_local <= 0 || stride <= 0) return MPMC_ERR_ARG;
	const int R = cm->n_ranks;
	std::vector<double> tmp((size_t)R * n_local * stride);
	int rc = mpmc_comm_allgather_f64(cm, local, (int64_t)n_local * stride, tmp.data());
	if (rc != MPMC_OK) return rc;
	for (int r = 0; r < R; r++)
		for (int slot = 0; slot < n_local; slot++)
			std::memcpy(all + ((size_t)slot * R + r) * stride, tmp.data() + ((size_t)r * n_local + slot) * stride, stride * sizeof(double));
	return MPMC_OK;
}

// ---- one process, G devices: PI_calculate_potential end to end --------------------------------------------------------------------
// One host thread per device, kept for the life of the process (the reference's OpenMP team, PathIntegral.cpp:772-779): creating and
// joining G threads inside every step would put ~0.1-0.2 ms of thread start-up in front of a 4 ms step at G = 8.
namespace {
struct DevWorker {
	std::thread th;
	std::mutex mu;
	std::condition_variable cv;
	std::function<void()> job;
	bool busy = false, stop = false;
	void loop() {
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			cv.wait(lk, [this] { return busy || stop; });
			if (stop) return;
			lk.unlock();
			job();
			lk.lock();
			busy = false;
			cv.notify_all();
		}
	}
	void post(std::function<void()> f) {
		std::lock_guard<std::mutex> lk(mu);
		job = std::move(f);
		busy = true;
		cv.notify_all();
	}
	void wait() {
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [this] { return !busy; });
	}
};
// the process-wide communicator of one set of devices and the host threads that serve devices 1 .. G-1 (device 0: the caller's thread)
struct AutoGroup {
	mpmc_comm *cm = nullptr;
	std::vector<std::unique_ptr<DevWorker>> workers;
	std::mutex step_mu; // one PI step at a time per set of devices
};
} // namespace
static std::mutex g_auto_mu;
static std::map<std::vector<int>, std::unique_ptr<AutoGroup>> g_auto_groups;
// A communicator that is still alive when the process ends takes RCCL's own teardown down with it (seen: "double free or corruption" at
// interpreter exit).  The handler is registered when the first one is made -- after RCCL was loaded, so it runs BEFORE RCCL's and HIP's
// static destructors.
static void destroy_auto_comms() {
	std::lock_guard<std::mutex> lk(g_auto_mu);
	for (auto &kv : g_auto_groups) {
		for (auto &w : kv.second->workers) {
			{
				std::lock_guard<std::mutex> wl(w->mu);
				w->stop = true;
				w->cv.notify_all();
			}
			if (w->th.joinable()) w->th.join();
		}
		mpmc_comm_destroy(kv.second->cm);
	}
	g_auto_groups.clear();
}

// devices of the beads in order of first appearance (rank g of the communicator = g-th device) and every bead's place.
// Test hook: a context configured with "virtual_device" = v >= 0 counts as living on device -(1 + v): a one-GPU box then drives the
// thread-per-device path for real (worker hand-off, per-thread evaluation, ordered combine) -- tools/host_tsan.sh, tests/test_gpu_comm.py --
// with a host copy in RCCL's place (RCCL admits one rank per physical device).
static int device_key(const mpmc_ctx *c) { return c->tune.virtual_device >= 0 ? -(1 + c->tune.virtual_device) : c->device; }
static void group_beads(mpmc_ctx **beads, int n_beads, std::vector<int> &devs, std::vector<int> &dev_of, std::vector<int> &slot_of,
                        std::vector<std::vector<int>> &members) {
	dev_of.assign(n_beads, 0);
	slot_of.assign(n_beads, 0);
	for (int b = 0; b < n_beads; b++) {
		int g = 0;
		for (; g < (int)devs.size(); g++)
			if (devs[g] == device_key(beads[b])) break;
		if (g == (int)devs.size()) {
			devs.push_back(device_key(beads[b]));
			members.emplace_back();
		}
		dev_of[b] = g;
		slot_of[b] = (int)members[g].size();
		members[g].push_back(b);
	}
}

// size of the process-wide communicator mpmc_pi_allreduce uses for these beads (0: none made yet), and how many devices they are on
extern "C" int mpmc_pi_allreduce_info(mpmc_ctx **beads, int n_beads, int *n_devices, int *comm_n_ranks) {
	if (!beads || n_beads <= 0) return MPMC_ERR_ARG;
	for (int b = 0; b < n_beads; b++)
		if (!beads[b]) return MPMC_ERR_ARG;
	std::vector<int> devs, dev_of, slot_of;
	std::vector<std::vector<int>> members;
	group_beads(beads, n_beads, devs, dev_of, slot_of, members);
	if (n_devices) *n_devices = (int)devs.size();
	if (comm_n_ranks) {
		std::lock_guard<std::mutex> lk(g_auto_mu);
		auto it = g_auto_groups.find(devs);
		*comm_n_ranks = it == g_auto_groups.end() ? 0 : it->second->cm->n_ranks;
	}
	return MPMC_OK;
}

extern "C" int mpmc_pi_allreduce(mpmc_ctx **beads, int n_beads, double sums4[4], mpmc_result *per_bead, int *any_failed) {
	if (!beads || n_beads <= 0 || !sums4) return MPMC_ERR_ARG;
	for (int b = 0; b < n_beads; b++)
		if (!beads[b]) return MPMC_ERR_ARG;
	std::vector<int> devs, dev_of, slot_of;
	std::vector<std::vector<int>> members;
	group_beads(beads, n_beads, devs, dev_of, slot_of, members);
	const int G = (int)devs.size();
	AutoGroup *grp = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_auto_mu);
		auto it = g_auto_groups.find(devs);
		if (it == g_auto_groups.end()) {
			mpmc_comm *cm = nullptr;
			bool any_virtual = false;
			for (int d : devs) any_virtual |= d < 0;
			if (any_virtual) { // (test hook: virtual devices of one GPU, no RCCL)
				cm = new mpmc_comm();
				cm->n_ranks = G, cm->rank0 = 0, cm->host_only = true;
				cm->members.resize(G);
				for (int g = 0; g < G; g++) cm->members[g].device = beads[members[g][0]]->device;
			} else {
				int rc = mpmc_comm_init_all(&cm, G, devs.data());
				if (rc != MPMC_OK) return fail(beads[0], rc, "mpmc_pi_allreduce: " + g_comm_error);
			}
			if (g_auto_groups.empty()) std::atexit(destroy_auto_comms);
			std::unique_ptr<AutoGroup> ng(new AutoGroup);
			ng->cm = cm;
			for (int g = 1; g < G; g++) {
				ng->workers.emplace_back(new DevWorker);
				DevWorker *w = ng->workers.back().get();
				w->th = std::thread([w] { w->loop(); });
			}
			grp = ng.get();
			g_auto_groups[devs] = std::move(ng);
		} else {
			grp = it->second.get();
		}
	}
	mpmc_comm *cm = grp->cm;
	std::lock_guard<std::mutex> step_lk(grp->step_mu);
	// evaluate: one host thread per device enqueues that device's beads (all before the first wait) and waits for them --
	// the reference's "#pragma omp parallel for" over the beads (PathIntegral.cpp:772-779), folded to one thread per GPU
	std::vector<mpmc_result> res(n_beads);
	std::vector<int> rcs(G, MPMC_OK);
	auto work = [&](int g) {
		std::vector<mpmc_ctx *> mine;
		for (int b : members[g]) mine.push_back(beads[b]);
		std::vector<mpmc_result> r(mine.size());
		double s4[4];
		int failed = 0;
		rcs[g] = mpmc_pi_potential_local(mine.data(), (int)mine.size(), s4, r.data(), &failed);
		for (size_t k = 0; k < mine.size(); k++) res[members[g][k]] = r[k];
	};
	for (int g = 1; g < G; g++) grp->workers[g - 1]->post([&work, g] { work(g); });
	work(0);
	for (int g = 1; g < G; g++) grp->workers[g - 1]->wait();
	for (int g = 0; g < G; g++)
		if (rcs[g] != MPMC_OK) return rcs[g];
	// combine: every device contributes {rd, coulombic, polarization, vdw, iterator_failed} of its beads, padded to the largest share
	size_t per = 0;
	for (auto &m : members) per = std::max(per, m.size());
	constexpr int kStride = 5;
	std::vector<std::vector<double>> send(G, std::vector<double>(per * kStride, 0.0));
	for (int b = 0; b < n_beads; b++) {
		double *o = send[dev_of[b]].data() + (size_t)slot_of[b] * kStride;
		o[0] = res[b].rd_energy;
		o[1] = res[b].coulombic_energy;
		o[2] = res[b].polarization_energy;
		o[3] = res[b].vdw_energy;
		o[4] = (double)res[b].iterator_failed;
	}
	std::vector<double> gathered(per * kStride * G);
	std::vector<const double *> loc(G);
	std::vector<double *> all(G, nullptr);
	for (int g = 0; g < G; g++) loc[g] = send[g].data();
	all[0] = gathered.data(); // every device receives the same bytes; the host reads them from the first
	{ // (step_mu: one collective at a time on a shared communicator)
		int rc = allgather_members(cm, loc, per * kStride, all);
		if (rc != MPMC_OK) return fail(beads[0], rc, "mpmc_pi_allreduce: " + cm->err);
	}
	sums4[0] = sums4[1] = sums4[2] = sums4[3] = 0;
	int failed = 0;
	for (int b = 0; b < n_beads; b++) { // ordered accumulation s = 0..P-1 of what came back over RCCL, PathIntegral.cpp:791-796
		const double *v = gathered.data() + ((size_t)dev_of[b] * per + slot_of[b]) * kStride;
		for (int k = 0; k < 4; k++) sums4[k] += v[k];
		failed |= (v[4] != 0.0);
		if (per_bead) per_bead[b] = res[b];
	}
	if (any_failed) *any_failed = failed;
	return MPMC_OK;
}
