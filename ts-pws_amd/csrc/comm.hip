// comm.hip -- trace shards on several devices of ONE process: the RCCL all-reduce of the stacks (single
// process, ncclCommInitAll, one stream per device, group calls) and the sharded whole call the drop-in tspws_main uses
// when TSPWS_DEVICES names more than one device.
//
// The sum that shards is partial_linear_stacks (/root/reference/src/ts_pws1f_lib.c:866-881): P[g] = sum of the traces of
// group g, g = floor(i Kmax / mtr) from the GLOBAL trace index.  Shard r holds the contiguous traces
// [r mtr / n, (r + 1) mtr / n), streams them into ITS P[Kmax][N] (rows of groups it does not touch are zero), ONE
// ncclAllReduce(sum, fp64) per device over xGMI adds the shards, then the finish stage (Kmax transforms, weights, two
// inverses) is split by scales over the devices and a second, small all-reduce (2 N doubles) adds the partial
// reconstructions.  Single-stage calls reduce ST || PS instead.
//
// RCCL is bound at run time (dlopen of librccl.so.1, prototypes from <rccl/rccl.h>): a single-GPU user never loads the
// half-gigabyte library, and inside a Python process that already carries torch's copy the loader hands back that one.
//
// "local" backend (TSPWS_COMM=local, or a device list that names a device twice -- RCCL refuses duplicates): every entry of
// the list must be the SAME device; the all-reduce is then this file's own kernel (rank order, deterministic).  It exists so
// that the N-way bookkeeping of the sharded call can be exercised on a one-GPU box; it is not a fallback, and a list with
// distinct devices is refused under it (comm_create): several physical devices always reduce through RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <thread>

#include "tspws_internal.h"

namespace {
struct Rccl {
	void *so = nullptr;
	decltype(&ncclCommInitAll) CommInitAll = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclReduce) Reduce = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	decltype(&ncclGetVersion) GetVersion = nullptr;
} g_rccl;

int rccl_load()
{
	if (g_rccl.so) return 0;
	const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	void *so = nullptr;
	for (const char *n : names) if ((so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
	if (!so) return fail(TSPWS_E_NODEV, "comm: librccl.so.1 not found (RCCL is needed for more than one device)");
#define BIND(f) do { g_rccl.f = (decltype(g_rccl.f))dlsym(so, "nccl" #f); if (!g_rccl.f) { dlclose(so); return fail(TSPWS_E_NODEV, "comm: librccl lacks nccl" #f); } } while (0)
	BIND(CommInitAll); BIND(CommDestroy); BIND(AllReduce); BIND(Reduce); BIND(GroupStart); BIND(GroupEnd); BIND(GetErrorString); BIND(GetVersion);
#undef BIND
	g_rccl.so = so;
	return 0;
}

int nccl_fail(const char *what, ncclResult_t r)
{
	std::string m = what;
	m += ": ";
	m += g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error";
	return fail(TSPWS_E_HIP, m.c_str());
}
#define NCCL_TRY(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) return nccl_fail(#expr, r_); } while (0)
} // namespace

struct tspws_hip_comm {
	int ndev = 0;
	bool local = false;                 // own reduction kernel instead of RCCL (test backend, see the header comment)
	std::vector<int> dev;
	std::vector<ncclComm_t> comms;
	std::vector<hipStream_t> streams;   // one per list entry
	std::vector<hipStream_t> rstreams;  // ... and a second one per entry: reductions that run beside the streaming of the next piece
	std::vector<hipEvent_t> ev;         // local backend: "buffer i is ready" / "sum is ready"
	std::vector<hipEvent_t> ev_a, ev_b; // hand-over between streams[i] and rstreams[i] (tspws_hip_multi_stack's pieces)
	hipEvent_t ev_sum = nullptr;
	double **d_ptrs = nullptr;          // local backend: device copy of the buffer pointers
};

// buf[0][i] = buf[0][i] + buf[1][i] + ... in list order, then copied to the others (local backend: all buffers on the one device)
__global__ void __launch_bounds__(256) k_local_allreduce(double *const *__restrict__ bufs, int n, size_t count)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= count) return;
	double a = bufs[0][i];
	for (int r = 1; r < n; r++) a += bufs[r][i];
	for (int r = 0; r < n; r++) bufs[r][i] = a;
}

extern "C" int tspws_hip_comm_create(tspws_hip_comm **out, int ndev, const int *devices)
{
	if (!out || ndev < 1) return fail(TSPWS_E_ARG, "comm_create: bad argument");
	*out = nullptr;
	const int have = tspws_hip_device_count();
	tspws_hip_comm *c = new (std::nothrow) tspws_hip_comm;
	if (!c) return fail(TSPWS_E_NOMEM, "comm_create: host allocation");
	c->ndev = ndev;
	bool dup = false;
	for (int i = 0; i < ndev; i++) {
		const int d = devices ? devices[i] : i;
		if (d < 0 || d >= have) { delete c; return fail(TSPWS_E_NODEV, "comm_create: no such HIP device"); }
		for (int j = 0; j < i; j++) dup |= c->dev[j] == d;
		c->dev.push_back(d);
	}
	const char *e = getenv("TSPWS_COMM");
	c->local = dup || (e && !strcmp(e, "local"));
	if (e && !strcmp(e, "rccl") && dup) { delete c; return fail(TSPWS_E_ARG, "comm_create: RCCL cannot take a device twice"); }
	// the local backend is a test vehicle for ONE physical device named several times: with distinct devices in the list the
	// sum has to go through RCCL / xGMI -- never through this file's own kernel and peer mappings
	bool distinct = false;
	for (int i = 1; i < ndev; i++) distinct |= c->dev[i] != c->dev[0];
	if (c->local && distinct) {
		delete c;
		return fail(TSPWS_E_ARG, "comm_create: the local backend (TSPWS_COMM=local / a repeated device) takes one physical device only; distinct devices reduce through RCCL");
	}
	for (int i = 0; i < ndev; i++) {
		hipStream_t s = nullptr, s2 = nullptr;
		hipEvent_t ev = nullptr, ea = nullptr, eb = nullptr;
		if (hipSetDevice(c->dev[i]) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess ||
		    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) {
			if (s) c->streams.push_back(s);
			if (s2) c->rstreams.push_back(s2);
			if (ev) c->ev.push_back(ev);
			if (ea) c->ev_a.push_back(ea);
			if (eb) c->ev_b.push_back(eb);
			tspws_hip_comm_destroy(c);
			return fail(TSPWS_E_HIP, "comm_create: streams");
		}
		c->streams.push_back(s);
		c->rstreams.push_back(s2);
		c->ev.push_back(ev);
		c->ev_a.push_back(ea);
		c->ev_b.push_back(eb);
	}
	if (c->local) {
		(void)hipSetDevice(c->dev[0]);
		if (hipEventCreateWithFlags(&c->ev_sum, hipEventDisableTiming) != hipSuccess || hipMalloc(&c->d_ptrs, ndev * sizeof(double *)) != hipSuccess) {
			tspws_hip_comm_destroy(c);
			return fail(TSPWS_E_HIP, "comm_create: local backend");
		}
	} else {
		int rc = rccl_load();
		if (rc) { tspws_hip_comm_destroy(c); return rc; }
		c->comms.assign(ndev, nullptr);
		const ncclResult_t r = g_rccl.CommInitAll(c->comms.data(), ndev, c->dev.data());
		if (r != ncclSuccess) { c->comms.clear(); tspws_hip_comm_destroy(c); return nccl_fail("ncclCommInitAll", r); }
	}
	*out = c;
	return 0;
}

extern "C" void tspws_hip_comm_destroy(tspws_hip_comm *c)
{
	if (!c) return;
	for (size_t i = 0; i < c->comms.size(); i++) if (c->comms[i]) (void)g_rccl.CommDestroy(c->comms[i]);
	for (size_t i = 0; i < c->streams.size(); i++) { (void)hipSetDevice(c->dev[i]); (void)hipStreamDestroy(c->streams[i]); }
	for (size_t i = 0; i < c->rstreams.size(); i++) { (void)hipSetDevice(c->dev[i]); (void)hipStreamDestroy(c->rstreams[i]); }
	for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
	for (hipEvent_t e : c->ev_a) (void)hipEventDestroy(e);
	for (hipEvent_t e : c->ev_b) (void)hipEventDestroy(e);
	if (c->ev_sum) (void)hipEventDestroy(c->ev_sum);
	if (c->d_ptrs) (void)hipFree(c->d_ptrs);
	delete c;
}

extern "C" int tspws_hip_comm_size(const tspws_hip_comm *c) { return c ? c->ndev : 0; }
extern "C" int tspws_hip_comm_device(const tspws_hip_comm *c, int i) { return (c && i >= 0 && i < c->ndev) ? c->dev[i] : -1; }
extern "C" void *tspws_hip_comm_stream(const tspws_hip_comm *c, int i) { return (c && i >= 0 && i < c->ndev) ? (void *)c->streams[i] : nullptr; }
extern "C" const char *tspws_hip_comm_backend(const tspws_hip_comm *c)
{
	static thread_local char buf[64];
	if (!c) return "";
	if (c->local) return "local";
	int v = 0;
	if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v);
	snprintf(buf, sizeof buf, "rccl %d.%d.%d", v / 10000, (v / 100) % 100, v % 100);
	return buf;
}

// In-place sum over the devices: d_bufs[i] holds `count` doubles on device i; the reduction of device i is ordered on
// streams[i] (NULL: the communicator's own stream of that device).
extern "C" int tspws_hip_allreduce_f64(tspws_hip_comm *c, double *const *d_bufs, size_t count, void *const *streams)
{
	if (!c || !d_bufs) return fail(TSPWS_E_ARG, "allreduce: NULL");
	if (!count) return 0;
	auto st = [&](int i) { return streams && streams[i] ? (hipStream_t)streams[i] : c->streams[i]; };
	if (!c->local) {
		if (c->ndev == 1 && !getenv("TSPWS_COMM")) return 0; // one device: the sum is the buffer itself (TSPWS_COMM=rccl still goes through RCCL)
		NCCL_TRY(g_rccl.GroupStart());
		for (int i = 0; i < c->ndev; i++) {
			const ncclResult_t r = g_rccl.AllReduce(d_bufs[i], d_bufs[i], count, ncclDouble, ncclSum, c->comms[i], st(i));
			if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return nccl_fail("ncclAllReduce", r); }
		}
		NCCL_TRY(g_rccl.GroupEnd());
		return 0;
	}
	// local backend: every stream signals its buffer, stream 0 adds them in list order and writes all of them, the others wait
	for (int i = 1; i < c->ndev; i++) {
		HIP_TRY(hipSetDevice(c->dev[i]));
		HIP_TRY(hipEventRecord(c->ev[i], st(i)));
	}
	HIP_TRY(hipSetDevice(c->dev[0]));
	for (int i = 1; i < c->ndev; i++) HIP_TRY(hipStreamWaitEvent(st(0), c->ev[i], 0));
	HIP_TRY(hipMemcpyAsync(c->d_ptrs, d_bufs, c->ndev * sizeof(double *), hipMemcpyHostToDevice, st(0)));
	HIP_TRY(hipStreamSynchronize(st(0))); // (the pointer table is caller memory)
	hipLaunchKernelGGL(k_local_allreduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st(0), (double *const *)c->d_ptrs, c->ndev, count);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(c->ev_sum, st(0)));
	for (int i = 1; i < c->ndev; i++) {
		HIP_TRY(hipSetDevice(c->dev[i]));
		HIP_TRY(hipStreamWaitEvent(st(i), c->ev_sum, 0));
	}
	return 0;
}

// buf[root][i] = buf[0][i] + buf[1][i] + ... in list order; the other buffers are left alone
__global__ void __launch_bounds__(256) k_local_reduce(double *const *__restrict__ bufs, int n, int root, size_t count)
{
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= count) return;
	double a = bufs[0][i];
	for (int r = 1; r < n; r++) a += bufs[r][i];
	bufs[root][i] = a;
}

// Sum over the devices into device `root` only (the replica rows of a jackknife: one owner finishes them): d_bufs[i] holds `count`
// doubles on device i; ordered on streams[i] like tspws_hip_allreduce_f64.  Several calls may be grouped by the caller's order only --
// each call is its own RCCL group.
extern "C" int tspws_hip_reduce_f64(tspws_hip_comm *c, double *const *d_bufs, size_t count, int root, void *const *streams)
{
	if (!c || !d_bufs || root < 0 || root >= c->ndev) return fail(TSPWS_E_ARG, "reduce: bad argument");
	if (!count) return 0;
	auto st = [&](int i) { return streams && streams[i] ? (hipStream_t)streams[i] : c->streams[i]; };
	if (!c->local) {
		if (c->ndev == 1 && !getenv("TSPWS_COMM")) return 0;
		NCCL_TRY(g_rccl.GroupStart());
		for (int i = 0; i < c->ndev; i++) {
			const ncclResult_t r = g_rccl.Reduce(d_bufs[i], d_bufs[i], count, ncclDouble, ncclSum, root, c->comms[i], st(i));
			if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return nccl_fail("ncclReduce", r); }
		}
		NCCL_TRY(g_rccl.GroupEnd());
		return 0;
	}
	// local backend: every stream signals its buffer, the root's stream adds them in list order
	for (int i = 0; i < c->ndev; i++) {
		if (i == root) continue;
		HIP_TRY(hipSetDevice(c->dev[i]));
		HIP_TRY(hipEventRecord(c->ev[i], st(i)));
	}
	HIP_TRY(hipSetDevice(c->dev[root]));
	for (int i = 0; i < c->ndev; i++) if (i != root) HIP_TRY(hipStreamWaitEvent(st(root), c->ev[i], 0));
	HIP_TRY(hipMemcpyAsync(c->d_ptrs, d_bufs, c->ndev * sizeof(double *), hipMemcpyHostToDevice, st(root)));
	HIP_TRY(hipStreamSynchronize(st(root))); // (the pointer table is caller memory)
	hipLaunchKernelGGL(k_local_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st(root), (double *const *)c->d_ptrs, c->ndev, root, count);
	HIP_TRY(hipGetLastError());
	// (the other streams may reuse their buffers only after the sum has read them)
	HIP_TRY(hipEventRecord(c->ev_sum, st(root)));
	for (int i = 0; i < c->ndev; i++) {
		if (i == root) continue;
		HIP_TRY(hipSetDevice(c->dev[i]));
		HIP_TRY(hipStreamWaitEvent(st(i), c->ev_sum, 0));
	}
	return 0;
}

// ------------------------------------------------------------------------------------------
// the sharded whole call
// ------------------------------------------------------------------------------------------
extern "C" void tspws_shard_range(size_t mtr, unsigned r, unsigned n, size_t *first, size_t *count)
{
	const size_t a = (size_t)r * mtr / n, b = (size_t)(r + 1) * mtr / n;
	if (first) *first = a;
	if (count) *count = b - a;
}

struct tspws_hip_multi {
	tspws_hip_comm *comm = nullptr;
	std::vector<tspws_hip_plan *> plans;
	std::vector<double *> x2;        // per device: partial reconstructions [2 N]
	std::vector<float *> d_shard;    // per device: its shard of the traces (grown on demand, kept between calls)
	std::vector<size_t> shard_bytes;
	float *d_out = nullptr;          // first device: ls | tsPWS
};

extern "C" void tspws_hip_multi_destroy(tspws_hip_multi *m)
{
	if (!m) return;
	for (size_t i = 0; i < m->plans.size(); i++) tspws_hip_plan_destroy(m->plans[i]);
	for (size_t i = 0; i < m->x2.size(); i++) if (m->x2[i]) { (void)hipSetDevice(tspws_hip_comm_device(m->comm, (int)i)); (void)hipFree(m->x2[i]); }
	for (size_t i = 0; i < m->d_shard.size(); i++) if (m->d_shard[i]) { (void)hipSetDevice(tspws_hip_comm_device(m->comm, (int)i)); (void)hipFree(m->d_shard[i]); }
	if (m->d_out) { (void)hipSetDevice(tspws_hip_comm_device(m->comm, 0)); (void)hipFree(m->d_out); }
	tspws_hip_comm_destroy(m->comm);
	delete m;
}

extern "C" int tspws_hip_multi_create(tspws_hip_multi **out, int ndev, const int *devices, int type, unsigned J, unsigned V, unsigned N, double s0,
                                      double b0, double w0, int uni)
{
	if (!out) return fail(TSPWS_E_ARG, "multi_create: NULL");
	*out = nullptr;
	tspws_hip_multi *m = new (std::nothrow) tspws_hip_multi;
	if (!m) return fail(TSPWS_E_NOMEM, "multi_create: host allocation");
	int rc = tspws_hip_comm_create(&m->comm, ndev, devices);
	if (rc) { delete m; return rc; }
	m->plans.assign(ndev, nullptr);
	m->x2.assign(ndev, nullptr);
	m->d_shard.assign(ndev, nullptr);
	m->shard_bytes.assign(ndev, 0);
	for (int i = 0; i < ndev && !rc; i++) {
		rc = tspws_hip_plan_create(&m->plans[i], type, J, V, N, s0, b0, w0, uni, m->comm->dev[i]);
		if (!rc && hipMalloc(&m->x2[i], 2 * (size_t)N * sizeof(double)) != hipSuccess) rc = fail(TSPWS_E_NOMEM, "multi_create: device allocation");
	}
	if (!rc && (hipSetDevice(m->comm->dev[0]) != hipSuccess || hipMalloc(&m->d_out, 2 * (size_t)N * sizeof(float)) != hipSuccess))
		rc = fail(TSPWS_E_NOMEM, "multi_create: device allocation");
	if (rc) { tspws_hip_multi_destroy(m); return rc; }
	*out = m;
	return 0;
}

// Host traces -> shards: the host array is pinned (portable: every device can pull from it) in page-aligned 128-MB pieces and every
// device copies ITS shard on its own stream, so the PCIe links of all devices run at the same time -- and every device starts as soon
// as the first piece of its shard is pinned (one blocking registration of the whole array in front of all copies was 10 ms / GB:
// 52 GB at BASELINE configs[4]).  The pieces are cut over the WHOLE array (two registrations must not share a page; a piece that
// straddles a shard boundary is registered once and copied in two parts) and handed out round-robin over the devices.
// *d_shards = the per-device buffers (owned by m, kept for the next call); *d_ls / *d_ts = the output buffers on the first device.
extern "C" int tspws_hip_multi_upload(tspws_hip_multi *m, const float *h_sigall, size_t ld, size_t mtr, const float *const **d_shards, float **d_ls,
                                      float **d_ts)
{
	if (!m || !h_sigall || !d_shards) return fail(TSPWS_E_ARG, "multi_upload: NULL");
	tspws_hip_comm *c = m->comm;
	const int n = c->ndev;
	const size_t total = mtr * ld * sizeof(float);
	int rc = 0;
	std::vector<uintptr_t> s0(n), s1(n); // host byte range of every shard
	for (int r = 0; r < n && !rc; r++) {
		size_t first, count;
		tspws_shard_range(mtr, (unsigned)r, (unsigned)n, &first, &count);
		const size_t bytes = count * ld * sizeof(float);
		s0[r] = (uintptr_t)(h_sigall + first * ld); s1[r] = s0[r] + bytes;
		if (hipSetDevice(c->dev[r]) != hipSuccess) { rc = fail(TSPWS_E_HIP, "multi_upload: device"); break; }
		if (m->shard_bytes[r] < bytes) {
			if (m->d_shard[r]) (void)hipFree(m->d_shard[r]);
			m->d_shard[r] = nullptr; m->shard_bytes[r] = 0;
			if (hipMalloc(&m->d_shard[r], std::max<size_t>(bytes, 16)) != hipSuccess) { rc = fail(TSPWS_E_NOMEM, "multi_upload: device allocation"); break; }
			m->shard_bytes[r] = bytes;
		}
	}
	if (rc) return rc;
	const size_t piece = (size_t)128 << 20;
	const uintptr_t h0 = (uintptr_t)h_sigall, h1 = h0 + total;
	const bool pin = total >= ((size_t)32 << 20);
	const size_t npieces = total ? (total + piece - 1) / piece : 0;
	auto cut = [&](size_t k) -> uintptr_t { if (k == 0) return h0; if (k >= npieces) return h1; return std::min((h0 + k * piece + 4095) & ~(uintptr_t)4095, h1); };
	std::vector<char> state(npieces, 0); // 0: untouched, 1: registered, 2: registration failed (travels pageable)
	std::vector<size_t> next(n);         // next piece of every shard
	for (int r = 0; r < n; r++) { size_t k = 0; while (k < npieces && cut(k + 1) <= s0[r]) k++; next[r] = k; }
	bool more = true;
	while (more && !rc) {
		more = false;
		for (int r = 0; r < n && !rc; r++) {
			const size_t k = next[r];
			if (k >= npieces || cut(k) >= s1[r] || s0[r] == s1[r]) continue;
			more = true;
			next[r] = k + 1;
			const uintptr_t a = cut(k), b = cut(k + 1);
			if (pin && state[k] == 0) {
				if (b > a && hipHostRegister((void *)a, b - a, hipHostRegisterPortable) == hipSuccess) state[k] = 1;
				else { (void)hipGetLastError(); state[k] = 2; }
			}
			const uintptr_t lo = std::max(a, s0[r]), hi = std::min(b, s1[r]);
			if (hi <= lo) continue;
			if (hipSetDevice(c->dev[r]) != hipSuccess ||
			    hipMemcpyAsync((char *)m->d_shard[r] + (lo - s0[r]), (const void *)lo, hi - lo, hipMemcpyHostToDevice, c->streams[r]) != hipSuccess)
				rc = fail(TSPWS_E_HIP, "multi_upload: copy");
		}
	}
	for (int r = 0; r < n; r++) { (void)hipSetDevice(c->dev[r]); if (hipStreamSynchronize(c->streams[r]) != hipSuccess && !rc) rc = fail(TSPWS_E_HIP, "multi_upload: sync"); }
	for (size_t k = 0; k < npieces; k++) if (state[k] == 1) (void)hipHostUnregister((void *)cut(k));
	if (rc) return rc;
	*d_shards = (const float *const *)m->d_shard.data();
	if (d_ls) *d_ls = m->d_out;
	if (d_ts) *d_ts = m->d_out + m->plans[0]->N;
	return 0;
}

// fold / mean removal on the shards (in place, float: ts_pws1f_lib.c:71-88, :159-169), mirrored back into the host array like
// the single-device call does
extern "C" int tspws_hip_multi_prologue(tspws_hip_multi *m, float *h_sigall, size_t max, size_t ld, size_t mtr, int fold, int rm)
{
	if (!m || !h_sigall) return fail(TSPWS_E_ARG, "multi_prologue: NULL");
	if (!fold && !rm) return 0;
	tspws_hip_comm *c = m->comm;
	const int n = c->ndev;
	int rc;
	for (int r = 0; r < n; r++) {
		size_t first, count;
		tspws_shard_range(mtr, (unsigned)r, (unsigned)n, &first, &count);
		if (!count) continue;
		HIP_TRY(hipSetDevice(c->dev[r]));
		if (fold && (rc = tspws_hip_fold(m->d_shard[r], count, max, ld, c->streams[r]))) return rc;
		if (rm && (rc = tspws_hip_remove_mean(m->d_shard[r], count, max, ld, c->streams[r]))) return rc;
		HIP_TRY(hipMemcpyAsync(h_sigall + first * ld, m->d_shard[r], count * ld * sizeof(float), hipMemcpyDeviceToHost, c->streams[r]));
	}
	for (int r = 0; r < n; r++) { HIP_TRY(hipSetDevice(c->dev[r])); HIP_TRY(hipStreamSynchronize(c->streams[r])); }
	return 0;
}

extern "C" tspws_hip_comm *tspws_hip_multi_comm(tspws_hip_multi *m) { return m ? m->comm : nullptr; }
extern "C" tspws_hip_plan *tspws_hip_multi_plan(tspws_hip_multi *m, int i) { return (m && i >= 0 && i < (int)m->plans.size()) ? m->plans[i] : nullptr; }

// First piece of the two-piece streaming / reduction schedule (the Python binding's split_groups): redundant finish -- about half the
// groups, the second reduction then hides behind the transforms of the first half; scale-sharded finish (nothing to hide behind) -- all
// but the last two groups; even when possible (the streaming pass launches the groups two at a time).
static unsigned multi_split_groups(unsigned K, bool sharded_finish)
{
	unsigned half = (sharded_finish && K > 3) ? K - 2 : K / 2;
	if (half >= 2 && (half & 1)) half--;
	return half;
}

// One tspws_main-equivalent call over trace shards that already sit on the devices: d_shards[r] = traces
// [first_r, first_r + count_r) of the ensemble (tspws_shard_range) on device r, row stride ld.  d_ls / d_ts (max floats each)
// live on device 0.  All work is ordered on the communicator's streams; returns after synchronising them.
// The ONE logical fp64 reduction of the stage-1 buffer is placed by TSPWS_SCHEDULE exactly as in ts-pws_amd.stack_sharded (one process
// per GPU): "single" = local halves, one all-reduce of the whole buffer, finish on the first device; "split" = the groups in two pieces,
// the first reduction (on the communicator's second streams) beside the streaming of the second piece, the second beside the
// transforms of the first piece; "sharded-finish" = pieces K - 2 | 2 + the finish stage split by scales over the devices.  Default: "single"
// (north_star's wording, and the one collective sequence that needs nothing but an all-reduce) -- the overlapped schedules issue reductions
// on the communicator from two streams per device and ncclReduce to owners, which have run over the local backend / one RCCL rank only;
// they are opt-in until a node with several GPUs has measured them (bench.py --gpus N times all three in one run).
extern "C" int tspws_hip_multi_stack(tspws_hip_multi *m, const t_tsPWS *p, const float *const *d_shards, size_t ld, size_t mtr, float *d_ls,
                                     float *d_ts)
{
	if (!m || !p || !d_shards || !d_ls || !d_ts || !mtr) return fail(TSPWS_E_ARG, "multi_stack: bad argument");
	tspws_hip_comm *c = m->comm;
	const int n = c->ndev;
	const size_t N = m->plans[0]->N;
	const unsigned K = p->Kmax;
	int rc;
	const char *se = getenv("TSPWS_SCHEDULE");
	int schedule = 0; // 0 single (default), 1 split, 2 sharded-finish
	if (se && *se) {
		if (!strcmp(se, "single")) schedule = 0; else if (!strcmp(se, "split")) schedule = 1; else if (!strcmp(se, "sharded-finish")) schedule = 2;
		else return fail(TSPWS_E_ARG, "multi_stack: TSPWS_SCHEDULE must be single, split or sharded-finish");
	}
	std::vector<double *> bufs(n);
	size_t nd = 0;
	for (int r = 0; r < n; r++) if ((rc = tspws_hip_reduce_buffer(m->plans[r], p, mtr, &bufs[r], &nd))) return rc;
	// finish stage by scales over the devices when every plan can (two-stage, polyphase kernels)
	std::vector<unsigned> s0(n), s1(n);
	bool sharded = n > 1 && schedule == 2;
	for (int r = 0; r < n && sharded; r++) {
		const int q = tspws_hip_finish_shard(m->plans[r], p, mtr, (unsigned)r, (unsigned)n, &s0[r], &s1[r]);
		if (q == 1) sharded = false; else if (q) return q;
	}
	const bool two_stage = tspws_is_two_stage(p, mtr);
	const bool pieces = n > 1 && schedule != 0 && two_stage && K >= 2;
	if (!pieces) {
		// local halves, one device after the other: every call below only enqueues work on that device's stream
		for (int r = 0; r < n; r++) {
			size_t first, count;
			tspws_shard_range(mtr, (unsigned)r, (unsigned)n, &first, &count);
			if ((rc = tspws_hip_stack_local(m->plans[r], p, d_shards[r], ld, count, first, mtr, c->streams[r]))) return rc;
		}
		if ((rc = tspws_hip_allreduce_f64(c, bufs.data(), nd, nullptr))) return rc;
		if (sharded) {
			for (int r = 0; r < n; r++)
				if ((rc = tspws_hip_stack_finish_scales(m->plans[r], p, mtr, s0[r], s1[r], m->x2[r], c->streams[r]))) return rc;
			if ((rc = tspws_hip_allreduce_f64(c, m->x2.data(), 2 * N, nullptr))) return rc;
			HIP_TRY(hipSetDevice(c->dev[0]));
			if ((rc = tspws_hip_epilogue(d_ls, d_ts, m->x2[0] + N, m->x2[0], N, (unsigned)mtr, c->streams[0]))) return rc;
		} else if ((rc = tspws_hip_stack_finish(m->plans[0], p, mtr, d_ls, d_ts, c->streams[0]))) return rc;
	} else {
		const unsigned half = std::max(1u, multi_split_groups(K, sharded));
		std::vector<void *> rs(n);
		std::vector<double *> b2(n);
		for (int r = 0; r < n; r++) { rs[r] = (void *)c->rstreams[r]; b2[r] = bufs[r] + (size_t)half * N; }
		auto stream_piece = [&](unsigned g0, unsigned g1) -> int {
			for (int r = 0; r < n; r++) {
				size_t first, count;
				tspws_shard_range(mtr, (unsigned)r, (unsigned)n, &first, &count);
				HIP_TRY(hipSetDevice(c->dev[r]));
				if (int q = tspws_hip_partial_stacks_range(m->plans[r], d_shards[r], ld, count, first, mtr, K, g0, g1, bufs[r], N, c->streams[r])) return q;
			}
			return 0;
		};
		auto hand_over = [&](std::vector<hipEvent_t> &ev, bool to_reduce) -> int { // streams[r] -> rstreams[r] (or back)
			for (int r = 0; r < n; r++) {
				HIP_TRY(hipSetDevice(c->dev[r]));
				HIP_TRY(hipEventRecord(ev[r], to_reduce ? c->streams[r] : c->rstreams[r]));
				HIP_TRY(hipStreamWaitEvent(to_reduce ? c->rstreams[r] : c->streams[r], ev[r], 0));
			}
			return 0;
		};
		// piece 1 streamed; its reduction on the second streams while piece 2 is streamed
		if ((rc = stream_piece(0, half))) return rc;
		if ((rc = hand_over(c->ev_a, true))) return rc;
		if ((rc = tspws_hip_allreduce_f64(c, bufs.data(), (size_t)half * N, rs.data()))) return rc;
		if ((rc = stream_piece(half, K))) return rc;
		if (sharded) {
			// both reductions, then every device transforms / weights / reconstructs its share of the scales
			if ((rc = hand_over(c->ev_b, true))) return rc;
			if ((rc = tspws_hip_allreduce_f64(c, b2.data(), (size_t)(K - half) * N, rs.data()))) return rc;
			if ((rc = hand_over(c->ev_a, false))) return rc;
			for (int r = 0; r < n; r++)
				if ((rc = tspws_hip_stack_finish_scales(m->plans[r], p, mtr, s0[r], s1[r], m->x2[r], c->streams[r]))) return rc;
			if ((rc = tspws_hip_allreduce_f64(c, m->x2.data(), 2 * N, nullptr))) return rc;
			HIP_TRY(hipSetDevice(c->dev[0]));
			if ((rc = tspws_hip_epilogue(d_ls, d_ts, m->x2[0] + N, m->x2[0], N, (unsigned)mtr, c->streams[0]))) return rc;
		} else {
			// the first device transforms the reduced first piece while the second piece is being reduced
			HIP_TRY(hipSetDevice(c->dev[0]));
			HIP_TRY(hipEventRecord(c->ev_a[0], c->rstreams[0])); // (first reduction done on device 0)
			if ((rc = hand_over(c->ev_b, true))) return rc;
			if ((rc = tspws_hip_allreduce_f64(c, b2.data(), (size_t)(K - half) * N, rs.data()))) return rc;
			HIP_TRY(hipSetDevice(c->dev[0]));
			HIP_TRY(hipStreamWaitEvent(c->streams[0], c->ev_a[0], 0));
			if ((rc = tspws_hip_stack_finish_range(m->plans[0], p, mtr, 0, half, c->streams[0]))) return rc;
			HIP_TRY(hipEventRecord(c->ev_b[0], c->rstreams[0])); // (second reduction done)
			HIP_TRY(hipStreamWaitEvent(c->streams[0], c->ev_b[0], 0));
			if ((rc = tspws_hip_stack_finish_range(m->plans[0], p, mtr, half, K, c->streams[0]))) return rc;
			if ((rc = tspws_hip_stack_finish_tail(m->plans[0], p, mtr, d_ls, d_ts, c->streams[0]))) return rc;
		}
	}
	for (int r = 0; r < n; r++) {
		HIP_TRY(hipSetDevice(c->dev[r]));
		HIP_TRY(hipStreamSynchronize(c->rstreams[r]));
		HIP_TRY(hipStreamSynchronize(c->streams[r]));
	}
	return 0;
}

// The stack and its C jackknife replicas over the shards: ONE pass per shard (tspws_hip_jackknife_local), the plain rows and
// the replicas' rows are all-reduced, device r finishes the contiguous block of replicas [r C / n, (r + 1) C / n) and hands
// its rows of the outputs straight to the HOST arrays h_ls_out / h_ts_out ([C][max] floats, row stride max); h_mtr_out
// receives the replica sizes.  h_sel = [C][mtr] selection over the whole ensemble.
extern "C" int tspws_hip_multi_stack_jackknife(tspws_hip_multi *m, const t_tsPWS *p, const float *const *d_shards, size_t ld, size_t mtr, float *d_ls,
                                               float *d_ts, const char *h_sel, unsigned C, float *h_ls_out, float *h_ts_out, unsigned *h_mtr_out)
{
	if (!m || !p || !d_shards || !d_ls || !d_ts || !mtr || !h_sel || !C || !h_ls_out || !h_ts_out || !h_mtr_out)
		return fail(TSPWS_E_ARG, "multi_stack_jackknife: bad argument");
	if (!tspws_is_two_stage(p, mtr)) return fail(TSPWS_E_ARG, "multi_stack_jackknife: two-stage calls only");
	tspws_hip_comm *c = m->comm;
	const int n = c->ndev;
	const size_t N = m->plans[0]->N;
	int rc;
	std::vector<double *> bufs(n), rows(n);
	size_t nd = 0, nr = 0;
	for (int r = 0; r < n; r++) {
		size_t first, count;
		tspws_shard_range(mtr, (unsigned)r, (unsigned)n, &first, &count);
		if ((rc = tspws_hip_jackknife_local(m->plans[r], p, d_shards[r], ld, count, first, mtr, h_sel, C, c->streams[r]))) return rc;
		if ((rc = tspws_hip_reduce_buffer(m->plans[r], p, mtr, &bufs[r], &nd))) return rc;
		if ((rc = tspws_hip_jackknife_buffer(m->plans[r], p, C, &rows[r], &nr))) return rc;
	}
	if ((rc = tspws_hip_allreduce_f64(c, bufs.data(), nd, nullptr))) return rc;
	// the replicas' rows go to their OWNER only (device r finishes the block [r C / n, (r + 1) C / n)): one reduction per owner instead
	// of an all-reduce of all C Kmax N doubles on every device -- as ts-pws_amd.jackknife_sharded does
	(void)nr;
	for (int r = 0; r < n; r++) {
		const unsigned c0 = (unsigned)((size_t)r * C / n), c1 = (unsigned)((size_t)(r + 1) * C / n);
		if (c1 == c0) continue;
		std::vector<double *> blk(n);
		for (int q = 0; q < n; q++) blk[q] = rows[q] + (size_t)c0 * p->Kmax * N;
		if ((rc = tspws_hip_reduce_f64(c, blk.data(), (size_t)(c1 - c0) * p->Kmax * N, r, nullptr))) return rc;
	}
	if ((rc = tspws_hip_stack_finish(m->plans[0], p, mtr, d_ls, d_ts, c->streams[0]))) return rc;
	// every device finishes ITS block of replicas at the same time: one host thread per device (the finish call synchronises
	// its stream, and the current device is a per-thread setting)
	std::vector<float *> d_rep(n, nullptr);
	std::vector<int> rcs(n, 0);
	std::vector<std::string> msgs(n);
	std::vector<std::thread> th;
	for (int r = 0; r < n; r++) {
		const unsigned c0 = (unsigned)((size_t)r * C / n), c1 = (unsigned)((size_t)(r + 1) * C / n);
		if (c1 == c0) continue;
		th.emplace_back([&, r, c0, c1]() {
			int q = 0;
			if (hipSetDevice(c->dev[r]) != hipSuccess || hipMalloc(&d_rep[r], 2 * (size_t)C * N * sizeof(float)) != hipSuccess)
				q = fail(TSPWS_E_NOMEM, "multi_stack_jackknife: device allocation");
			// (jackknife_finish addresses rows c0.. of [C][N] arrays and synchronises its stream)
			if (!q) q = tspws_hip_jackknife_finish(m->plans[r], p, mtr, h_sel, C, c0, c1, d_rep[r], d_rep[r] + (size_t)C * N, h_mtr_out, c->streams[r]);
			if (!q && (hipMemcpy(h_ls_out + (size_t)c0 * N, d_rep[r] + (size_t)c0 * N, (size_t)(c1 - c0) * N * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
			           hipMemcpy(h_ts_out + (size_t)c0 * N, d_rep[r] + (size_t)(C + c0) * N, (size_t)(c1 - c0) * N * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess))
				q = fail(TSPWS_E_HIP, "multi_stack_jackknife: download");
			rcs[r] = q;
			if (q) msgs[r] = tspws_hip_last_error(); // (the error text is per thread)
		});
	}
	for (std::thread &t : th) t.join();
	for (int r = 0; r < n; r++) if (rcs[r] && !rc) rc = fail(rcs[r], msgs[r].c_str());
	for (int r = 0; r < n; r++) if (d_rep[r]) { (void)hipSetDevice(c->dev[r]); (void)hipFree(d_rep[r]); }
	if (rc) return rc;
	for (int r = 0; r < n; r++) { HIP_TRY(hipSetDevice(c->dev[r])); HIP_TRY(hipStreamSynchronize(c->streams[r])); }
	return 0;
}
