// libbenchpush_hip.so: C ABI (include/benchpush_amd.h) over the gfx950 kernels.  No torch types, no CPU fallback:
// every entry point fails with BP_ENODEVICE / BP_EHIP when the GPU path cannot run.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include "bp_host_bd.hpp"
#include "bp_boxdelivery.hpp"
#include "bp_policy.hpp"

struct bp_handle {
    bp_config cfg;
    int num_envs = 0;
    long long env_offset = 0;
    int device = 0;
    int nbcap = 0;
    int num_trials = 0;
    bool loaded = false, was_reset = false;
    int *order_buf = nullptr;
    bool steps_done = false;
    bool resettle = false; // true: reset() re-runs the settle sub-steps instead of copying the settled template
    bool maze8 = false;    // maze whose hulls all have <= 8 vertices: kernels instantiated with 8-vertex loops
    bool damp = false;     // bp_config.damping_pow != 0: k_physics_step_damp / k_physics_reset_damp (generic vertex loops, no scheduler)
    int pair_mode = 0;              // two envs per wavefront (bp_physics_pair.hpp): 1 = fixed pairs for the whole step (k_physics_step_pair), 2 = inside the scheduler
    int pair_resident = 0;          // > 0: pairing launches run k_physics_step_schedr with this many resident workgroups (BP_PAIR_RESIDENT)
    int sched_persist = 0;          // > 0: the scheduled launch is k_physics_step_schedl with this many resident workgroups (BP_SCHED_PERSIST)
    void *pd_buf = nullptr;         // its launch constants in device memory (DevParams, DevPtrs)
    int sched_chunk = 0;            // > 0: k_physics_step_sched (preemptive scheduler, chunks of this many sub-steps) is the step kernel; BP_SCHED=0 turns it off
    std::string dev_lock_key;       // non-empty: this handle holds a share of the per-device residency lock (resident_* below)
    bool resident_auto = false;     // resident launches fall back to dispatcher-driven ones while another PROCESS uses the device (checked every few launches)
    bool device_shared = false;     // result of the last check
    unsigned launches = 0;
    int num_cus = 0;                // multiProcessorCount of the handle's device
    hipStream_t st_aux = nullptr;   // box-delivery / area-clearing: the robot's spfa map runs beside the finish kernel; ship-ice: the solo kernel of a pairing launch
    std::vector<hipStream_t> st_parts;   // ship-ice: the scheduled launch split over several hardware queues (a measured variant; P.sq_parts is 1 since round 6)
    std::vector<hipEvent_t> ev_parts;
    hipStream_t st_aux2 = nullptr;  // box-delivery / area-clearing: pass 1 of the two-pass step (the envs that ran out of pass 0's sim-step budget) and its tail kernels
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
    hipStream_t st_aux3 = nullptr;  // ... pass 1's kernel with the recurrence test of execute_robot_path, beside the lean one
    hipEvent_t ev_join3 = nullptr;
    int bd_budget = 0;              // sim steps of pass 0 of the two-pass step, 0 = one pass (BP_BD_BUDGET)
    DevParams P;
    DevPtrs D;
    std::vector<void *> allocs;
    size_t lds_bytes = 0, obs_lds_bytes = 0;
    size_t sched_lds = 0;           // dynamic LDS of the scheduler kernels: lds_bytes, or two half-wave images in pairing launches if that is more
    std::string err;
    std::vector<double> goal_raw; // maze: un-normalised wavefront map (info['goal_dt'])
    // box-delivery
    bp_bd_config bdcfg;
    BdParams B;
    BdPtrs Q;
    size_t bd_rmap_lds = 0;
    size_t bd_lds = 0, bd_obs_lds = 0;
    std::vector<bpgeom::BdMaps> bd_maps;
    std::vector<int> bd_map_of_trial;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev; // triples: start, mid, stop
    size_t ev_used = 0;
    unsigned long long *cost_ring = nullptr;   // [BP_COST_RING][2] per-launch (sum, max) of the envs' step cycles while timing is on (k_cost_stats)
    int cost_n = 0;
};
#define BP_COST_RING 1024

static int fail(bp_handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    return code;
}
#define HIPCHK(h, call)                                                                                       \
    do {                                                                                                      \
        hipError_t _e = (call);                                                                               \
        if (_e != hipSuccess) return fail((h), BP_EHIP, std::string(#call) + ": " + hipGetErrorString(_e));    \
    } while (0)

// Entry points run on the handle's device and leave the calling thread's current device as they found it.
struct DevGuard {
    int prev = -1;
    bool changed = false, ok = true;
    explicit DevGuard(int d)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != d) { ok = hipSetDevice(d) == hipSuccess; changed = ok; }
    }
    ~DevGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
};
#define BP_DEVICE(h) DevGuard _dg((h)->device); if (!_dg.ok) return fail((h), BP_EHIP, "hipSetDevice failed")

template <typename T>
static int dalloc(bp_handle *h, T **p, size_t n, int fill_byte = 0)
{
    void *q = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(h, BP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemset(q, fill_byte, bytes);
    if (e != hipSuccess) return fail(h, BP_EHIP, std::string("hipMemset: ") + hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = (T *)q;
    return BP_OK;
}

static int mvcap_for(int nbcap) { return nbcap > 192 ? nbcap : 192; }
#ifdef BP_PROF
#define BP_PROF_HOST true
#else
#define BP_PROF_HOST false
#endif
static size_t lds_bytes_for(int nbcap, bool box) { return bp_lds_map(nbcap, mvcap_for(nbcap), box, BP_PROF_HOST).total; }

// The chunk boundaries at which a running env may yield (bit k: after k chunks).  A launch without pairing takes no turns during the first 40 % of the step: the
// even finish needs fine turns only towards the end, and every turn costs ~76 us of slot-time -- +1.6 % env-steps/s at 4096 envs, +1.8 % maze, +0.8 %
// at 50 % concentration; pairing launches (-2 % at 5120 envs) keep every boundary.  BP_SCHED_YMASK=<mask> overrides.
static unsigned sched_yield_mask(bool solo_launch, int steps, int chunk)
{
    if (const char *ev = getenv("BP_SCHED_YMASK")) return (unsigned)strtoul(ev, nullptr, 0);
    if (!solo_launch || chunk <= 0) return 0xFFFFFFFFu;
    const int first = (2 * steps + 5 * chunk - 1) / (5 * chunk);   // ceil(0.4 * steps / chunk)
    return first >= 32 ? 0u : (0xFFFFFFFFu << first);
}
// ---- is this process alone on its device? --------------------------------------------------------------------------------------------------------
// A resident kernel holds every wave slot until its launch is over: two PROCESSES that share one GPU can then only alternate by saving and restoring
// 2 048 wavefronts (80 ms per launch measured in a two-rank rehearsal on one device).  One rank per GPU -- the deployment this library is
// built for -- never sees that, but nothing used to detect the other case.  Every handle with resident launches now holds a SHARED advisory lock on a
// per-device file (keyed by the PCI bus id, so that HIP_VISIBLE_DEVICES renumbering does not matter), one descriptor per process and device; "alone" =
// the lock can be converted to EXCLUSIVE without waiting.  Checked at load and every 64 launches (one flock call); while it fails the handle launches the
// dispatcher-driven kernels (bit-identical results, BP_SCHED_PERSIST=0 behaviour) and goes back to resident ones when the other process has left.
// BP_SCHED_PERSIST / BP_PAIR_RESIDENT set explicitly switch the detection off.  Best effort: processes that do not share a /tmp are not seen.
struct ResidentLock { int fd = -1; int refs = 0; };
static std::mutex g_res_mu;
static std::map<std::string, ResidentLock> g_res_locks;

static void resident_acquire(bp_handle *h)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), h->device) != hipSuccess) return;
    for (char *c = bus; *c; c++) if (*c == ':' || *c == '/' || *c == '.') *c = '_';
    const char *dir = getenv("BP_LOCK_DIR");
    const std::string key = std::string(dir ? dir : "/tmp") + "/benchpush_amd.resident." + bus + ".lock";
    std::lock_guard<std::mutex> g(g_res_mu);
    ResidentLock &L = g_res_locks[key];
    if (L.fd < 0) {
        const mode_t um = umask(0);
        L.fd = open(key.c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
        umask(um);
        if (L.fd < 0) { g_res_locks.erase(key); return; }   // no lock file: assume the device is ours
        (void)flock(L.fd, LOCK_SH);
    }
    L.refs++;
    h->dev_lock_key = key;
}
static void resident_release(bp_handle *h)
{
    if (h->dev_lock_key.empty()) return;
    std::lock_guard<std::mutex> g(g_res_mu);
    auto it = g_res_locks.find(h->dev_lock_key);
    if (it != g_res_locks.end() && --it->second.refs <= 0) { close(it->second.fd); g_res_locks.erase(it); }
    h->dev_lock_key.clear();
}
// true: some other process holds a share of this device's lock
static bool resident_device_shared(bp_handle *h)
{
    if (h->dev_lock_key.empty()) return false;
    std::lock_guard<std::mutex> g(g_res_mu);
    auto it = g_res_locks.find(h->dev_lock_key);
    if (it == g_res_locks.end()) return false;
    if (flock(it->second.fd, LOCK_EX | LOCK_NB) == 0) { (void)flock(it->second.fd, LOCK_SH); return false; }
    (void)flock(it->second.fd, LOCK_SH);   // a failed conversion may have dropped the shared lock: take it again
    return true;
}
static int device_cus(bp_handle *h)
{
    if (h->num_cus <= 0) {
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, h->device));
        h->num_cus = prop.multiProcessorCount;
    }
    return BP_OK;
}
// Resident wavefronts for a scheduled launch without pairing (k_physics_step_schedl*): one workgroup per wave slot of the device, BP_SCHED_PERSIST=0 goes back to
// one workgroup per task from the hardware dispatcher, BP_SCHED_PERSIST=n > 1 launches n workgroups per slot (the surplus waits for the end and leaves)
static int sched_persist_setup(bp_handle *h)
{
    int rc = device_cus(h);
    if (rc) return rc;
    const int slots = h->num_cus * 4 * (h->P.env_kind == BP_ENV_SHIP_ICE ? BP_SCHED_WAVES : 2);   // 4 SIMDs x 2 wavefronts of 256 VGPRs (x 3 in the occupancy experiment)
    const int pers = getenv("BP_SCHED_PERSIST") ? atoi(getenv("BP_SCHED_PERSIST")) : 1;
    h->sched_persist = pers > 0 ? std::min(h->num_envs, slots * pers) : 0;
    if (h->sched_persist && !h->pd_buf) HIPCHK(h, hipMalloc(&h->pd_buf, sizeof(DevParams) + sizeof(DevPtrs)));
    if (h->sched_persist && !getenv("BP_SCHED_PERSIST")) {
        h->resident_auto = true;
        resident_acquire(h);
        h->device_shared = resident_device_shared(h);
    }
    return BP_OK;
}
extern "C" {

int32_t bp_abi_version(void) { return BP_ABI_VERSION; }
int32_t bp_sizeof_config(void) { return (int32_t)sizeof(bp_config); }

const char *bp_last_error(const bp_handle *h) { return h ? h->err.c_str() : "null handle"; }

int bp_create(const bp_config *cfg, int32_t num_envs, int64_t env_id_offset, int32_t device, bp_handle **out)
{
    if (!cfg || !out || num_envs <= 0) return BP_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BP_ENODEVICE;
    if (!(cfg->damping_pow >= 0.0) || cfg->damping_pow > 1.0) return BP_EINVAL;   // pow(space.damping, dt) of a damping in [0, 1]; != 0 selects the generic kernels (k_physics_*_damp)
    if (cfg->num_ship_verts < 3 || cfg->num_ship_verts > BP_MAX_SHIP_VERTS) return BP_EINVAL;
    if (cfg->steps <= 0 || cfg->iterations <= 0 || cfg->persistence <= 0) return BP_EINVAL;
    if (cfg->env_kind != BP_ENV_SHIP_ICE && cfg->env_kind != BP_ENV_MAZE) return BP_EINVAL;
    if (cfg->env_kind == BP_ENV_MAZE && (cfg->num_wheels < 0 || cfg->num_wheels > BP_MAX_WHEELS)) return BP_EINVAL;
    if (cfg->random_start && cfg->env_kind != BP_ENV_SHIP_ICE) return BP_EINVAL; // per-episode start poses: ship-ice only
    bp_handle *h = new bp_handle();
    h->cfg = *cfg;
    h->num_envs = num_envs;
    h->env_offset = env_id_offset;
    h->device = device;
    h->resettle = (cfg->env_kind == BP_ENV_SHIP_ICE && cfg->random_start != 0); // the settled field depends on the start pose
    DevGuard _dg(device);
    if (!_dg.ok) { delete h; return BP_ENODEVICE; }
    memset(&h->D, 0, sizeof(h->D));
    DevParams &P = h->P;
    memset(&P, 0, sizeof(P));
    P.dt_sub = cfg->dt / cfg->steps;
    P.steps = cfg->steps; P.iterations = cfg->iterations; P.persistence = cfg->persistence; P.settle_steps = cfg->settle_steps;
    if (const char *evp = getenv("BP_DEBUG_PATHS")) P.dbg_paths = atoi(evp); // test hook: force the rarely taken paths of the narrow phase (bp_device.hpp)
    P.damping_pow = cfg->damping_pow; P.bias_coef = cfg->bias_coef; P.slop = cfg->slop;
    P.target_speed = cfg->target_speed; P.max_yaw_rate = cfg->max_yaw_rate;
    P.map_w = cfg->map_w; P.map_h = cfg->map_h; P.goal_y = cfg->goal_y; P.m_to_pix = cfg->m_to_pix;
    P.beta = cfg->beta; P.boundary_penalty = cfg->boundary_penalty; P.terminal_reward = cfg->terminal_reward;
    P.local_w = cfg->local_w; P.local_h = cfg->local_h; P.vshift = cfg->vshift; P.obs_range = cfg->obs_range;
    P.skin = 0.06;  // Verlet skin of the neighbour lists (m): a tuning knob only, the lists are an exact broadphase for any value
    P.env_kind = cfg->env_kind;
    P.nkin = (cfg->env_kind == BP_ENV_MAZE) ? 1 + cfg->num_wheels : 1;
    P.goal_x = cfg->goal_x; P.goal_reach = cfg->goal_reach; P.k_increment = cfg->k_increment;
    P.num_envs = num_envs; P.env_offset = env_id_offset;
    P.random_start = (cfg->env_kind == BP_ENV_SHIP_ICE) ? cfg->random_start : 0;
    P.start_x_range = cfg->start_x_range; P.start_seed = cfg->start_seed; P.ship_mass = cfg->ship_mass;
    P.num_ship_verts = cfg->num_ship_verts;
    memcpy(P.ship_verts, cfg->ship_verts, sizeof(P.ship_verts));
    memcpy(P.ship_head, cfg->ship_head, sizeof(P.ship_head));
    memcpy(P.ship_tail, cfg->ship_tail, sizeof(P.ship_tail));
    // OccupancyGrid.__init__ (occupancy_map.py:11-35) with grid cell 1/m_to_pix
    const double grid = 1.0 / cfg->m_to_pix;
    P.grid_w = (int)(cfg->map_w / grid);
    P.grid_h = (int)(cfg->map_h / grid);
    P.obs_h = (int)(cfg->local_h * cfg->m_to_pix);
    P.obs_w = (int)(cfg->local_w * cfg->m_to_pix);
    if ((P.obs_h * P.obs_w) % 4 != 0) { delete h; return BP_EINVAL; }
    if (cfg->env_kind == BP_ENV_MAZE) {
        const int infl = (P.obs_w > P.obs_h ? P.obs_w : P.obs_h) / 2;
        const int nwords = ((P.obs_h + infl) * (P.obs_w + infl) + 31) / 32;
        h->obs_lds_bytes = (size_t)4 * (nwords + ((nwords + 1) & ~1)) + sizeof(double) * 2 * MZ_MAXBOX * 4;
        if (hipFuncSetAttribute((const void *)k_observe_maze, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->obs_lds_bytes) != hipSuccess) { delete h; return BP_EHIP; }
    } else {
        // k_observe composes two window rows as obs_w / 2 whole words per thread group
        if (P.obs_h % 2 != 0 || P.obs_w % 2 != 0 || P.obs_w / 2 > OBS_THREADS_SHIP) { delete h; return BP_EINVAL; }
        h->obs_lds_bytes = (size_t)((P.obs_h * P.obs_w + 15) & ~15) + (size_t)((P.obs_h + 15) & ~15) + sizeof(double) * 2 * OBS_CHUNK * BP_MAXV;
        if (hipFuncSetAttribute((const void *)k_observe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->obs_lds_bytes) != hipSuccess) { delete h; return BP_EHIP; }
    }
    *out = h;
    return BP_OK;
}

int bp_destroy(bp_handle *h)
{
    if (!h) return BP_EINVAL;
    DevGuard _dg(h->device);
    if (h->st_aux) { hipStreamSynchronize(h->st_aux); hipStreamDestroy(h->st_aux); }
    resident_release(h);
    if (h->pd_buf) hipFree(h->pd_buf);
    if (h->st_aux2) { hipStreamSynchronize(h->st_aux2); hipStreamDestroy(h->st_aux2); }
    if (h->st_aux3) { hipStreamSynchronize(h->st_aux3); hipStreamDestroy(h->st_aux3); }
    if (h->ev_join3) hipEventDestroy(h->ev_join3);
    for (hipStream_t s_ : h->st_parts) { hipStreamSynchronize(s_); hipStreamDestroy(s_); }
    for (hipEvent_t e_ : h->ev_parts) hipEventDestroy(e_);
    if (h->ev_join2) hipEventDestroy(h->ev_join2);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    for (void *p : h->allocs) hipFree(p);
    for (hipEvent_t e : h->ev) hipEventDestroy(e);
    delete h;
    return BP_OK;
}

// Common tail of the scenario loaders: SoA upload of the per-trial bodies, per-env state allocation, and one settle of
// every trial into its reset template (state slot num_envs + t).
static int upload_trials(bp_handle *h, const std::vector<std::vector<bpgeom::Shape>> &trials, bool settle = true)
{
    using namespace bpgeom;
    const int T = (int)trials.size();
    int maxnb = 1;
    for (const auto &b : trials) maxnb = std::max(maxnb, (int)b.size());
    if (maxnb > 60000) return fail(h, BP_EINVAL, "too many bodies");
    const int nbcap = (maxnb + 7) / 8 * 8;
    h->nbcap = nbcap;
    h->num_trials = T;
    h->P.nbcap = nbcap;
    h->P.mvcap = mvcap_for(nbcap);
    h->P.num_trials = T;
    h->lds_bytes = lds_bytes_for(nbcap, h->P.env_kind == BP_ENV_BOX);
    if (h->lds_bytes > 160 * 1024) return fail(h, BP_EINVAL, "nb_cap too large for LDS");
    // the candidate cache (LdsCtx::cc, bp_physics.hpp step 3) packs both body indices of a pair into BP_CC_IDX_BITS bits each
    if (nbcap >= (1 << BP_CC_IDX_BITS)) return fail(h, BP_EINVAL, "nb_cap exceeds the candidate cache's body-index field (16384 bodies per env)");

    std::vector<int> h_nb(T), h_nv((size_t)T * nbcap, 0), h_kind((size_t)T * nbcap, 0);
    std::vector<d2> h_lv((size_t)T * nbcap * BP_MAXV), h_ln((size_t)T * nbcap * BP_MAXV);
    std::vector<double4> h_mass((size_t)T * nbcap), h_pose((size_t)T * nbcap), h_prop((size_t)T * nbcap);
    memset(h_lv.data(), 0, h_lv.size() * sizeof(d2));
    memset(h_ln.data(), 0, h_ln.size() * sizeof(d2));
    memset(h_mass.data(), 0, h_mass.size() * sizeof(double4));
    memset(h_pose.data(), 0, h_pose.size() * sizeof(double4));
    memset(h_prop.data(), 0, h_prop.size() * sizeof(double4));
    for (int t = 0; t < T; t++) {
        h_nb[t] = (int)trials[t].size();
        for (int b = 0; b < (int)trials[t].size(); b++) {
            const Shape &s = trials[t][b];
            if ((int)s.verts.size() > BP_MAXV) return fail(h, BP_EINVAL, "hull exceeds BP_MAXV vertices");
            const size_t o = (size_t)t * nbcap + b;
            h_nv[o] = (int)s.verts.size();
            for (int i = 0; i < (int)s.verts.size(); i++) {
                h_lv[o * BP_MAXV + i].x = s.verts[i].x; h_lv[o * BP_MAXV + i].y = s.verts[i].y;
                h_ln[o * BP_MAXV + i].x = s.normals[i].x; h_ln[o * BP_MAXV + i].y = s.normals[i].y;
            }
            h_mass[o].x = s.m_inv; h_mass[o].y = s.i_inv; h_mass[o].z = s.cog.x; h_mass[o].w = s.cog.y;
            h_pose[o].x = s.p.x; h_pose[o].y = s.p.y; h_pose[o].z = s.angle; h_pose[o].w = 0.0;
            h_prop[o].x = s.radius; h_prop[o].y = s.e; h_prop[o].z = s.u; h_prop[o].w = 0.0;
            h_kind[o] = s.kind;
        }
    }
    DevPtrs &D = h->D;
    int rc;
    int *d_nb, *d_nv, *d_kind; d2 *d_lv, *d_ln; double4 *d_mass, *d_pose, *d_prop;
    if ((rc = dalloc(h, &d_nb, T))) return rc;
    if ((rc = dalloc(h, &d_nv, (size_t)T * nbcap))) return rc;
    if ((rc = dalloc(h, &d_kind, (size_t)T * nbcap))) return rc;
    if ((rc = dalloc(h, &d_lv, (size_t)T * nbcap * BP_MAXV))) return rc;
    if ((rc = dalloc(h, &d_ln, (size_t)T * nbcap * BP_MAXV))) return rc;
    if ((rc = dalloc(h, &d_mass, (size_t)T * nbcap))) return rc;
    if ((rc = dalloc(h, &d_pose, (size_t)T * nbcap))) return rc;
    if ((rc = dalloc(h, &d_prop, (size_t)T * nbcap))) return rc;
    HIPCHK(h, hipMemcpy(d_nb, h_nb.data(), sizeof(int) * T, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_nv, h_nv.data(), sizeof(int) * h_nv.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_kind, h_kind.data(), sizeof(int) * h_kind.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_lv, h_lv.data(), sizeof(d2) * h_lv.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_ln, h_ln.data(), sizeof(d2) * h_ln.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_mass, h_mass.data(), sizeof(double4) * h_mass.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_pose, h_pose.data(), sizeof(double4) * h_pose.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_prop, h_prop.data(), sizeof(double4) * h_prop.size(), hipMemcpyHostToDevice));
    D.sc_nb = d_nb; D.sc_nv = d_nv; D.sc_kind = d_kind; D.sc_lv = d_lv; D.sc_ln = d_ln; D.sc_mass = d_mass; D.sc_pose = d_pose;
    D.sc_prop = d_prop;

    // per-env state for the E envs plus T settled reset templates (slot E + t = trial t)
    const size_t E = (size_t)h->num_envs + (size_t)T, EB = E * nbcap;
    if ((rc = dalloc(h, &D.e_trial, E))) return rc;
    if ((rc = dalloc(h, &D.e_episode, E, 0xFF))) return rc; // -1
    if ((rc = dalloc(h, &D.e_nb, E))) return rc;
    if ((rc = dalloc(h, &D.e_err, E))) return rc;
    if ((rc = dalloc(h, &D.e_flags, E))) return rc;
    if ((rc = dalloc(h, &D.e_prevdist, E))) return rc;
    if ((rc = dalloc(h, &D.e_stamp, E))) return rc;
    if ((rc = dalloc(h, &D.e_currdt, E))) return rc;
    if ((rc = dalloc(h, &D.e_total_work, E))) return rc;
    if ((rc = dalloc(h, &D.e_ke, E))) return rc;
    if ((rc = dalloc(h, &D.e_imp, E))) return rc;
    if ((rc = dalloc(h, &D.e_cnt, E * 4))) return rc;
    if ((rc = dalloc(h, &D.e_cost, E))) return rc;
    if ((rc = dalloc(h, &h->order_buf, E))) return rc;
    D.order = nullptr;
    if ((rc = dalloc(h, &D.pxy, EB))) return rc;
    if ((rc = dalloc(h, &D.ang, EB))) return rc;
    if ((rc = dalloc(h, &D.rot, EB))) return rc;
    if ((rc = dalloc(h, &D.velv, EB))) return rc;
    if ((rc = dalloc(h, &D.velw, EB))) return rc;
    if ((rc = dalloc(h, &D.velb, EB))) return rc;
    if ((rc = dalloc(h, &D.wv, EB * BP_MAXV))) return rc;
    if ((rc = dalloc(h, &D.wn, EB * BP_MAXV))) return rc;
    if ((rc = dalloc(h, &D.pv, EB * BP_MAXV))) return rc;
    if ((rc = dalloc(h, &D.bb, EB))) return rc;
    if ((rc = dalloc(h, &D.fat, EB))) return rc;
    if ((rc = dalloc(h, &D.adj, EB * BP_KADJ))) return rc;
    if ((rc = dalloc(h, &D.adjn, EB))) return rc;
    if ((rc = dalloc(h, &D.hint, EB * BP_KADJ))) return rc;
    if ((rc = dalloc(h, &D.a_key, E * BP_ACAP, 0xFF))) return rc;
    if ((rc = dalloc(h, &D.a_stamp, E * BP_ACAP))) return rc;
    if ((rc = dalloc(h, &D.a_sc, E * BP_ACAP))) return rc;
    if ((rc = dalloc(h, &D.a_h0, E * BP_ACAP))) return rc;
    if ((rc = dalloc(h, &D.a_h1, E * BP_ACAP))) return rc;
    if ((rc = dalloc(h, &D.a_d, E * BP_ACAP * 14))) return rc;
    if ((rc = dalloc(h, &D.e_lastrew, E))) return rc;
    if ((rc = dalloc(h, &D.e_lastflag, E))) return rc;
    if ((rc = dalloc(h, &D.m_acc, E * 8))) return rc;
    if ((rc = dalloc(h, &D.m_rows, E * BP_EPM_COUNT))) return rc;
    if ((rc = dalloc(h, &D.m_ring, E * BP_EPM_RING * BP_EPM_COUNT))) return rc;
    if ((rc = dalloc(h, &D.m_sum, E * BP_EPM_COUNT))) return rc;
    if ((rc = dalloc(h, &D.m_count, E))) return rc;
    if ((rc = dalloc(h, &D.m_open, E))) return rc;
    if ((rc = dalloc(h, &D.clk, (size_t)16))) return rc;
    HIPCHK(h, hipMemset(D.clk, 0, 16 * sizeof(unsigned long long)));
    D.dbg = nullptr; D.dbg_env = -1; D.prof = nullptr;
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_reset, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    h->damp = h->P.damping_pow != 0.0;
    if (h->damp) {
        // space.damping != 0: one generic pair of kernels for ship-ice and maze handles; settle every trial once with it
        HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_step_damp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_reset_damp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        if (!settle) return BP_OK;
        hipLaunchKernelGGL(k_physics_reset_damp, dim3(T), dim3(64), h->lds_bytes, 0, h->P, h->D, (const unsigned char *)nullptr, (double *)nullptr, 1);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipDeviceSynchronize());
        h->loaded = true;
        return BP_OK;
    }
    if (h->P.env_kind == BP_ENV_SHIP_ICE && h->P.nkin == 1 && nbcap < 16384) {
        bool plain = true; // one kinematic shape (index 0), dynamic shapes without groups otherwise
        for (int t = 0; t < T && plain; t++)
            for (int b = 0; b < (int)trials[t].size(); b++) {
                const int kd = trials[t][b].kind;
                const int bt = (kd >> 16) & 3, grp = (kd >> 8) & 0xFF;
                if (grp != 0 || bt != (b == 0 ? 1 : 0)) { plain = false; break; }
            }
        // Preemptive scheduler (k_physics_step_sched), the default step kernel of a ship-ice handle: chunks of 40 sub-steps.  It pays while the launch
        // is a few rounds of the wave slots (+14 % at 4096 envs; -1 % at 16 384, where the tail is amortised): BP_SCHED=<chunk> forces it, BP_SCHED=0
        // selects the one-wave-per-env kernel.
        // two environments per wavefront: the half-wave LDS image is laid out for PP_NBCAP body slots, element offsets are 32-bit
        const bool can_pair = plain && nbcap <= PP_NBCAP &&
                              ((size_t)h->num_envs + (size_t)T) * (size_t)nbcap * BP_MAXV * sizeof(d2) < (size_t)0xFFFFFFFF &&   // 32-bit byte offsets (gA)
                              ((size_t)h->num_envs + (size_t)T) * (size_t)nbcap * BP_KADJ * sizeof(unsigned long long) < (size_t)0xFFFFFFFF;
        // Default: inside the scheduler (mode 2) from 5 120 envs per GPU up, where a launch is throughput (+3.7 % at 5 120, +12 % at 6 144, +19 % at 8 192 and
        // +31 % at 16 384 env-steps/s over the resident solo scheduler, same box).  Below, the launch follows the chains of its heaviest envs and the scheduler's rotation, not the sum of the work: pairing
        // the light envs saves a tenth of the wave-instructions and moves the launch by 0 ... +2.5 % at 4 096 envs (profiles/r05_pair/), -2 % at 2 048 --
        // and the kernel that holds both step bodies runs the solo body 6 % slower -- so those handles keep the lean one-env-per-wavefront scheduler kernel.
        // BP_PAIR=0 / 1 / 2 overrides.
        if ((rc = device_cus(h))) return rc;
        // regime boundaries in rounds of the device's wave slots, not in envs (bp_policy.hpp): pairing from 2.5 rounds, loose limits from 3.5, no scheduler
        // without pairing above 4 -- 5 120 / 7 168 / 8 192 envs on the 256 CUs they were measured on
        const BpLaunchPolicy pol = bp_launch_policy(h->num_envs, h->num_cus, can_pair, false);
        h->pair_mode = pol.pair_mode;
        if (const char *evp = getenv("BP_PAIR")) h->pair_mode = can_pair ? atoi(evp) : 0;
        if (h->pair_mode == 1) {
            h->P.pair_mode = 1;
            if (const char *evs = getenv("BP_PAIR_SNAKE")) h->P.pair_solo = atoi(evs);
            HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_step_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * PL_HALF)));
        }
        // (with paired first tasks the scheduler is what lets an env leave its pair, so it stays on at every batch size: chunks of 100 sub-steps above 8192 envs)
        int ch = pol.chunk ? pol.chunk : (h->pair_mode == 2 ? 100 : 0);
        if (const char *ev = getenv("BP_SCHED")) ch = atoi(ev);
        if (plain && ch > 0 && (h->P.steps + ch - 1) / ch <= SQ_MAXLEV && h->num_envs < (1 << 24)) {
            h->sched_chunk = ch;
            h->P.sq_chunk = ch; h->P.sq_levels = (h->P.steps + ch - 1) / ch;
            // Measured-and-lost scheduler variants (longest-remaining-first keys, hold, static class sets, the launch split over hardware queues: tools/experiments/README.md)
            // are no longer load-time switches (round 6).  Their kernel code paths stay compiled -- taking them out of physics_body / sched_body moved the register allocation
            // of k_physics_step_schedl and cost 1.0 % at 4096 envs, same box, both alternations (tools/experiments/r06_removed_scheduler_variants.diff) -- and are switched off here.
            h->P.sq_hold = 0;
            h->P.sq_lrpt = 0;
            h->P.sq_bw = 9000;
            h->P.sq_hyst = 1;
            h->P.sq_floor = 150;
            // pace-based issue priorities (physics_body: pace_prio), per cent of the reference cost for priority 3: +1.5 ... +2.4 % at 4096 envs, flat from 100 to 130
            h->P.sq_dynprio = getenv("BP_SCHED_DYNPRIO") ? atoi(getenv("BP_SCHED_DYNPRIO")) : 115;
            h->P.sq_ymask = 0xFFFFFFFFu;   // (set below, once it is known whether the launch pairs)
            h->P.sq_cls = 0;
            h->P.sq_parts = 1;
            h->P.sq_part = 0;
            h->P.sq_cap = h->num_envs; // an env's home XCD is where its first chunk ran: any share of the envs
            int *d_items, *d_ctr; unsigned *d_carry; unsigned char *d_moved;
            if ((rc = dalloc(h, &d_items, (size_t)SQ_NX * SQ_MAXLEV * h->P.sq_cap))) return rc;
            if ((rc = dalloc(h, &d_ctr, (size_t)SQ_NX * (SQ_MAXLEV + 2) * 2))) return rc;
            if ((rc = dalloc(h, &d_carry, (size_t)h->num_envs * 4))) return rc;
            if ((rc = dalloc(h, &d_moved, (size_t)h->num_envs * nbcap))) return rc;
            h->D.sq_items = d_items; h->D.sq_ctr = d_ctr; h->D.sq_carry = d_carry; h->D.sq_moved = d_moved;
            { int *d_done, *d_lev, *d_warn, *d_resc;
              if ((rc = dalloc(h, &d_resc, (size_t)1 + SQ_RESCUE))) return rc;
              h->D.sq_rescue = d_resc;
              if ((rc = dalloc(h, &d_done, (size_t)h->num_envs))) return rc;
              if ((rc = dalloc(h, &d_lev, (size_t)h->num_envs))) return rc;
              if ((rc = dalloc(h, &d_warn, (size_t)2))) return rc;
              HIPCHK(h, hipMemset(d_warn, 0, 2 * sizeof(int)));
              h->D.sq_done = d_done; h->D.sq_lev = d_lev; h->D.sq_warn = d_warn;
              unsigned *d_thr;
              if ((rc = dalloc(h, &d_thr, (size_t)1))) return rc;
              HIPCHK(h, hipMemset(d_thr, 0xFF, sizeof(unsigned)));
              h->D.sq_thr = d_thr;
              int *d_sub, *d_pst;
              if ((rc = dalloc(h, &d_sub, (size_t)h->num_envs))) return rc;
              if ((rc = dalloc(h, &d_pst, (size_t)8))) return rc;
              HIPCHK(h, hipMemset(d_pst, 0, 8 * sizeof(int)));
              h->D.sq_sub = d_sub; h->D.sq_pairstat = d_pst; }
#ifdef BP_DEBUG_PATHS   // fault injection lives in the diagnostic twin only: a stray variable in a job environment cannot disturb the product library
            if (const char *ev2 = getenv("BP_SCHED_DEBUG_DROP")) h->P.sq_debug = atoi(ev2);
#endif
            if (h->pair_mode == 2) {
                // two environments per wavefront inside the scheduler: who starts alone, and when a half leaves its pair (pair_should_leave)
                auto envint = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
                h->P.pair_mode = 2;
                // (pace priorities stay on: +0.4 ... +0.9 % in pairing launches)
                // Up to ~6 000 envs per GPU the launch is within a few per cent of the chain of its heaviest env: only envs that are light right now run
                // paired (a medium env beside a mate would become the longest chain), and the eighth of the dispatch order that was heaviest in the previous
                // step starts alone.  From ~7 000 envs the launch is throughput: every env starts in a pair and leaves it at 20 active arbiters or 40 work units per
                // sub-step (same-box sweeps in profiles/r05_pair/ and profiles/r05_sched/pairing_limits_by_batch.txt; on the resident kernel, fresh / steady state:
                // 6144 envs tight 294 k / 306 k against 294 k / 285 k, 7168 envs 300 k / 308 k against 314 k / 310 k, 8192 envs 303 k / 313 k against 323 k / 324 k).
                const bool tight = pol.pair_mode == 2 ? pol.tight != 0 : 2LL * h->num_envs < (long long)BP_POLICY_LOOSE_FROM_HALF_ROUNDS * pol.wave_slots;   // (BP_PAIR=2 forced below the default regime)
                h->P.pair_solo = std::min(h->num_envs, std::max(0, envint("BP_PAIR_SOLO", tight ? h->num_envs / 8 : 0)));
                h->P.pp_max_keys = std::min(30, envint("BP_PP_KEYS", 26));
                h->P.pp_max_slots = std::min(PP_NSLOT - 4, envint("BP_PP_SLOTS", 34));
                h->P.pp_max_mv = std::min(PP_MVCAP - 4, envint("BP_PP_MV", 40));
                h->P.pp_max_act = envint("BP_PP_ACT", tight ? 16 : 20);
                h->P.pp_max_work = envint("BP_PP_WORK", tight ? 9 : 40);
                h->P.pp_rate = envint("BP_PP_RATE", tight ? 70 : 200);
                h->P.pp_snake = envint("BP_PP_SNAKE", 0);
                h->P.pp_heavy_only = envint("BP_PP_HEAVY_ONLY", 0);
            }
            // (the per-env kernels k_physics_step / k_physics_reset keep lds_bytes, for which their attribute was set above)
            h->sched_lds = h->pair_mode == 2 ? std::max(h->lds_bytes, (size_t)(2 * PL_HALF)) : h->lds_bytes;
            HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_step_sched, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->sched_lds));
            HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_step_schedp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->sched_lds));
            HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_step_schedl, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->sched_lds));
            // (pairing launches have a resident kernel of their own, k_physics_step_schedr: the kernel that holds both step bodies INLINE lost 4 % at 8192 envs as a
            // resident loop -- 266 spilled VGPRs against 203)
            h->P.sq_ymask = sched_yield_mask(h->P.pair_mode != 2, h->P.steps, h->P.sq_chunk);
            if (h->P.pair_mode != 2 && h->P.sq_parts == 1) { int rc2 = sched_persist_setup(h); if (rc2) return rc2; }
            if (h->P.pair_mode == 2) {
                HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_step_schedr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->sched_lds));
                const int pr = getenv("BP_PAIR_RESIDENT") ? atoi(getenv("BP_PAIR_RESIDENT")) : 1;   // 0: the dispatcher-driven pair of kernels
                if (pr > 0) {
                    h->pair_resident = std::min(h->num_envs, pol.wave_slots * pr);
                    if (!h->pd_buf) HIPCHK(h, hipMalloc(&h->pd_buf, sizeof(DevParams) + sizeof(DevPtrs)));
                    if (!getenv("BP_PAIR_RESIDENT")) { h->resident_auto = true; resident_acquire(h); h->device_shared = resident_device_shared(h); }
                }
            }
        }
    }
    if (h->pair_mode == 2 && h->P.pair_mode != 2) h->pair_mode = 0;   // pairing inside the scheduler needs the scheduler
    if (!settle) return BP_OK;
    h->maze8 = (h->P.env_kind == BP_ENV_MAZE);
    for (int v : h_nv) if (v > 8) h->maze8 = false;
    if (h->maze8) {
        if ((rc = device_cus(h))) return rc;
        int ch = bp_launch_policy(h->num_envs, h->num_cus, false, true).chunk;
        if (const char *ev = getenv("BP_SCHED")) ch = atoi(ev);
        if (ch > 0 && (h->P.steps + ch - 1) / ch <= SQ_MAXLEV && h->num_envs < (1 << 24)) { // preemptive scheduler, as for ship-ice
            h->sched_chunk = ch;
            h->P.sq_chunk = ch; h->P.sq_levels = (h->P.steps + ch - 1) / ch; h->P.sq_cap = h->num_envs;
            h->P.sq_dynprio = getenv("BP_SCHED_DYNPRIO") ? atoi(getenv("BP_SCHED_DYNPRIO")) : 115;   // pace priorities as for ship-ice: +1.6 % at 4096 envs
            h->P.sq_ymask = sched_yield_mask(true, h->P.steps, h->P.sq_chunk);
            int *d_items, *d_ctr; unsigned *d_carry; unsigned char *d_moved;
            if ((rc = dalloc(h, &d_items, (size_t)SQ_NX * SQ_MAXLEV * h->P.sq_cap))) return rc;
            if ((rc = dalloc(h, &d_ctr, (size_t)SQ_NX * (SQ_MAXLEV + 2) * 2))) return rc;
            if ((rc = dalloc(h, &d_carry, (size_t)h->num_envs * 4))) return rc;
            if ((rc = dalloc(h, &d_moved, (size_t)h->num_envs * nbcap))) return rc;
            h->D.sq_items = d_items; h->D.sq_ctr = d_ctr; h->D.sq_carry = d_carry; h->D.sq_moved = d_moved;
            { int *d_done, *d_lev, *d_warn, *d_resc;
              if ((rc = dalloc(h, &d_resc, (size_t)1 + SQ_RESCUE))) return rc;
              h->D.sq_rescue = d_resc;
              if ((rc = dalloc(h, &d_done, (size_t)h->num_envs))) return rc;
              if ((rc = dalloc(h, &d_lev, (size_t)h->num_envs))) return rc;
              if ((rc = dalloc(h, &d_warn, (size_t)2))) return rc;
              HIPCHK(h, hipMemset(d_warn, 0, 2 * sizeof(int)));
              h->D.sq_done = d_done; h->D.sq_lev = d_lev; h->D.sq_warn = d_warn;
              unsigned *d_thr;
              if ((rc = dalloc(h, &d_thr, (size_t)1))) return rc;
              HIPCHK(h, hipMemset(d_thr, 0xFF, sizeof(unsigned)));
              h->D.sq_thr = d_thr;
              int *d_sub;
              if ((rc = dalloc(h, &d_sub, (size_t)h->num_envs))) return rc;
              h->D.sq_sub = d_sub; }
#ifdef BP_DEBUG_PATHS
            if (const char *ev2 = getenv("BP_SCHED_DEBUG_DROP")) h->P.sq_debug = atoi(ev2);
#endif
            h->sched_lds = h->lds_bytes;
            HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_step_sched_maze, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
            HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_step_schedl_maze, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
            if ((rc = sched_persist_setup(h))) return rc;
        }
        HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_step_maze, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        HIPCHK(h, hipFuncSetAttribute((const void *)k_physics_reset_maze, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
        hipLaunchKernelGGL(k_physics_reset_maze, dim3(T), dim3(64), h->lds_bytes, 0, h->P, h->D, (const unsigned char *)nullptr, (double *)nullptr, 1);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipDeviceSynchronize());
        h->loaded = true;
        return BP_OK;
    }
    // settle every trial once (new space + bodies + 1000 sub-steps, ship_ice_env.py:109-220); reset() copies from these
    hipLaunchKernelGGL(k_physics_reset, dim3(T), dim3(64), h->lds_bytes, 0, h->P, h->D, (const unsigned char *)nullptr, (double *)nullptr, 1);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipDeviceSynchronize());
    h->loaded = true;
    return BP_OK;
}

static int kind_of(int ctype, int group, int btype) { return ctype | (group << 8) | (btype << 16); }

int bp_load_scenarios(bp_handle *h, int32_t T, int32_t F, int32_t V, const double *verts, const int32_t *counts,
                      const double *centres, const double *starts, const int32_t *nfloes)
{
    if (!h || T <= 0 || F < 0 || V <= 0 || !starts || !nfloes) return BP_EINVAL;
    if (h->loaded) return fail(h, BP_ESTATE, "scenarios already loaded");
    if (h->P.env_kind != BP_ENV_SHIP_ICE) return fail(h, BP_ESTATE, "handle was created for another environment");
    BP_DEVICE(h);
    using namespace bpgeom;
    std::vector<std::vector<Shape>> trials(T);
    for (int t = 0; t < T; t++) {
        std::vector<Shape> &bodies = trials[t];
        Shape ship;
        build_ship(h->cfg.ship_verts, h->cfg.num_ship_verts, starts[3 * t], starts[3 * t + 1], starts[3 * t + 2], ship);
        ship.radius = h->cfg.poly_radius; ship.e = h->cfg.elasticity; ship.u = h->cfg.friction;
        ship.kind = kind_of(1, 0, BODY_KINEMATIC); // ship_ice_env.py:214-215
        bodies.push_back(ship);
        if (nfloes[t] > F) return fail(h, BP_EINVAL, "nfloes > F");
        for (int f = 0; f < nfloes[t]; f++) {
            const int n = counts[(size_t)t * F + f];
            if (n < 3) continue;
            if (n > V) return fail(h, BP_EINVAL, "vertex count > V");
            Shape s;
            if (!build_floe(verts + ((size_t)t * F + f) * V * 2, n, centres[((size_t)t * F + f) * 2],
                            centres[((size_t)t * F + f) * 2 + 1], h->cfg.density, h->cfg.poly_radius, s))
                continue;
            s.radius = h->cfg.poly_radius; s.e = h->cfg.elasticity; s.u = h->cfg.friction;
            s.kind = kind_of(2, 0, BODY_DYNAMIC);  // ship_ice_env.py:211-212
            bodies.push_back(s);
        }
    }
    return upload_trials(h, trials);
}

int bp_load_maze(bp_handle *h, int32_t T, int32_t nbox, const double *centres, int32_t nwalls, const double *walls, const double *start)
{
    if (!h || T <= 0 || nbox < 0 || nwalls < 0 || nwalls > 16 || !walls || !start || (nbox > 0 && !centres)) return BP_EINVAL;
    if (h->loaded) return fail(h, BP_ESTATE, "scenarios already loaded");
    if (h->P.env_kind != BP_ENV_MAZE) return fail(h, BP_ESTATE, "handle was created for another environment");
    BP_DEVICE(h);
    using namespace bpgeom;
    const bp_config &cf = h->cfg;
    std::vector<std::vector<Shape>> trials(T);
    for (int t = 0; t < T; t++) {
        std::vector<Shape> &bodies = trials[t];
        // robot: one KINEMATIC body with the main outline (type 1) + the wheels (type 0); friction stays at pymunk's
        // default 0 (Robot.sim sets mass and elasticity only, robot.py:90-105)
        for (int k = 0; k <= cf.num_wheels; k++) {
            Shape s;
            const double *st3 = start + 3 * (size_t)t;   // start pose of this layout (maze_NAMO_env.py:229-245)
            if (k == 0) build_kinematic_part(cf.ship_verts, cf.num_ship_verts, st3[0], st3[1], st3[2], s);
            else build_kinematic_part(cf.wheel_verts[k - 1], 4, st3[0], st3[1], st3[2], s);
            s.radius = cf.poly_radius; s.e = cf.elasticity; s.u = 0.0;
            s.kind = kind_of(k == 0 ? 1 : 0, 1, BODY_KINEMATIC);
            bodies.push_back(s);
        }
        for (int b = 0; b < nbox; b++) {
            const double ox = centres[((size_t)t * nbox + b) * 2], oy = centres[((size_t)t * nbox + b) * 2 + 1], sz = cf.obstacle_size;
            const double raw[8] = {ox + sz, oy + sz, ox - sz, oy + sz, ox - sz, oy - sz, ox + sz, oy - sz}; // maze_NAMO_env.py:317-320
            Shape s;
            if (!build_floe(raw, 4, ox, oy, cf.density, cf.poly_radius, s)) continue;
            s.radius = cf.poly_radius; s.e = cf.elasticity; s.u = cf.friction;
            s.kind = kind_of(2, 0, BODY_DYNAMIC);
            bodies.push_back(s);
        }
        for (int w = 0; w < nwalls; w++) {
            Shape s;
            build_wall(walls[4 * w], walls[4 * w + 1], walls[4 * w + 2], walls[4 * w + 3], s);
            s.radius = cf.wall_radius; s.e = 0.5; s.u = 0.5;            // sim_utils.py:177-180
            s.kind = kind_of(3, 0, BODY_STATIC);
            bodies.push_back(s);
        }
    }
    // static maps: wall raster + BFS goal map
    std::vector<unsigned char> wall;
    std::vector<double> norm;
    maze_maps(walls, nwalls, cf.wall_radius, cf.map_w, cf.map_h, h->P.grid_h, h->P.grid_w, cf.goal_x, cf.goal_y, wall, norm, h->goal_raw);
    double *d_norm; unsigned char *d_wall;
    int rc;
    if ((rc = dalloc(h, &d_norm, norm.size()))) return rc;
    if ((rc = dalloc(h, &d_wall, wall.size()))) return rc;
    HIPCHK(h, hipMemcpy(d_norm, norm.data(), sizeof(double) * norm.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(d_wall, wall.data(), wall.size(), hipMemcpyHostToDevice));
    h->D.dist_map = d_norm; h->D.wall_map = d_wall;
    {   // the observer's packed copy: sign = wall, magnitude = normalised distance (a wall cell's distance is exactly 1.0, never a zero)
        std::vector<double> packed(norm.size());
        for (size_t i = 0; i < norm.size(); i++) packed[i] = wall[i] ? -norm[i] : norm[i];
        double *d_pk;
        if ((rc = dalloc(h, &d_pk, packed.size()))) return rc;
        HIPCHK(h, hipMemcpy(d_pk, packed.data(), sizeof(double) * packed.size(), hipMemcpyHostToDevice));
        h->D.maze_obs_map = d_pk;
    }
    double *d_raw;
    if ((rc = dalloc(h, &d_raw, h->goal_raw.size()))) return rc;
    HIPCHK(h, hipMemcpy(d_raw, h->goal_raw.data(), sizeof(double) * h->goal_raw.size(), hipMemcpyHostToDevice));
    h->D.goal_raw = d_raw;
    return upload_trials(h, trials);
}

int bp_get_goal_map(bp_handle *h, double *out_host, int32_t *grid_h, int32_t *grid_w)
{
    if (!h) return BP_EINVAL;
    if (h->goal_raw.empty()) return fail(h, BP_ESTATE, "no goal map (not a maze handle or not loaded)");
    if (out_host) memcpy(out_host, h->goal_raw.data(), sizeof(double) * h->goal_raw.size());
    if (grid_h) *grid_h = h->P.grid_h;
    if (grid_w) *grid_w = h->P.grid_w;
    return BP_OK;
}

// python's round(x, 2) (ship_ice_env.py:337-339 rounds info['state']): the double nearest to x rounded to two decimals, ties of the
// exact value to even.  x * 100 is taken exactly (product + fma residual), so a product that only *rounds* onto a tie is not one.
__device__ __forceinline__ double bp_round2(double x)
{
    const double y = x * 100.0;
    const double e = __builtin_fma(x, 100.0, -y);
    double k = __builtin_rint(y);
    if (__builtin_fabs(y - k) == 0.5 && e != 0.0) { const double lo = __builtin_floor(y); k = (e > 0.0) ? lo + 1.0 : lo; }
    return k / 100.0;
}
__global__ __launch_bounds__(256) void k_debug_round2(const double *__restrict__ in, double *__restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bp_round2(in[i]);
}

// ShipIceMetric.reset / update (ship_ice_metric.py:26-69; shared arithmetic in benchpush_amd/metrics/interactive_nav.py) for every env,
// one thread per env, after the physics kernel of bp_step (mode 0) or the reset kernel of bp_reset (mode 1).
__global__ __launch_bounds__(256) void k_episode_metrics(const DevParams P, const DevPtrs D, const int mode, const unsigned char *__restrict__ mask)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    // bp_get_clock_stamps: the shader-clock counters of the XCDs are not synchronised with each other, so the pair is filed under the XCD that took it
    if (env == 0) { const int x = sq_xcc_id(); D.clk[2 * x] = __builtin_amdgcn_s_memtime(); D.clk[2 * x + 1] = __builtin_amdgcn_s_memrealtime(); }
    if (env >= P.num_envs) return;
    double *a = D.m_acc + (size_t)env * 8;
    const d2 p = D.pxy[(size_t)env * P.nbcap];
    const double x = bp_round2(p.x), y = bp_round2(p.y);
    auto emit = [&](double success) {
        const double l0 = a[1], L = a[4], work = a[6];
        const double own = P.ship_mass * l0;
        double *r = D.m_rows + (size_t)env * BP_EPM_COUNT;
        r[BP_EPM_EFFICIENCY] = (success != 0.0) ? L / l0 : 0.0;   // compute_efficiency_score
        r[BP_EPM_EFFORT] = own / (own + work);                    // compute_effort_score
        r[BP_EPM_REWARD] = a[0]; r[BP_EPM_SUCCESS] = success; r[BP_EPM_LENGTH] = a[5]; r[BP_EPM_TOTAL_WORK] = work;
        // the lists of BaseMetric (base_metric.py:12-16): ring of the last episodes and running sums over all of them
        const unsigned n = D.m_count[env];
        double *ring = D.m_ring + ((size_t)env * BP_EPM_RING + n % BP_EPM_RING) * BP_EPM_COUNT, *sum = D.m_sum + (size_t)env * BP_EPM_COUNT;
        for (int q = 0; q < BP_EPM_COUNT; q++) { ring[q] = r[q]; sum[q] += r[q]; }
        D.m_count[env] = n + 1u;
    };
    if (mode == 1) {
        if (mask != nullptr && mask[env] == 0) return;
        if (D.m_open[env] && a[5] > 0.0) emit(0.0);   // reset of a running episode: eps_complete by truncation
        double L = P.goal_y - y;                      // ShipIceMetric: goal line - start y (ship_ice_metric.py:52)
        if (P.env_kind == BP_ENV_MAZE) {              // MazeNamoMetric: wavefront length at the start pixel (maze_namo_metric.py:68-75)
            const int px = (int)(x * P.m_to_pix), py = (int)(y * P.m_to_pix);
            L = (px >= 0 && px < P.grid_w && py >= 0 && py < P.grid_h) ? D.goal_raw[(size_t)py * P.grid_w + px] / P.m_to_pix : 0.0;
        }
        a[0] = 0.0; a[1] = 0.0; a[2] = x; a[3] = y; a[4] = L; a[5] = 0.0; a[6] = 0.0; a[7] = 0.0;
        D.m_open[env] = 1;
        return;
    }
    if (!D.m_open[env]) return; // stepped past its end without a reset: nothing to account
    const int fl = D.e_lastflag[env];
    const double dx = a[2] - x, dy = a[3] - y;
    a[0] += D.e_lastrew[env];
    a[1] += __builtin_sqrt(dx * dx + dy * dy);
    a[2] = x; a[3] = y;
    a[5] += 1.0;
    a[6] = D.e_total_work[env];
    a[7] = (double)((fl >> 1) & 1);
    if (fl & 1) { emit(a[7]); D.m_open[env] = 0; }
}

static int launch(bp_handle *h, int mode, const double *actions, const unsigned char *mask, unsigned char *obs, double *reward,
                  unsigned char *term, unsigned char *trunc, double *info, hipStream_t st, bool physics, bool raster)
{
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    if (h->timing) {
        if (h->ev_used + 3 > h->ev.size()) {
            for (int k = 0; k < 3; k++) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); h->ev.push_back(e); }
        }
        e0 = h->ev[h->ev_used]; e1 = h->ev[h->ev_used + 1]; e2 = h->ev[h->ev_used + 2];
        h->ev_used += 3;
        HIPCHK(h, hipEventRecord(e0, st));
    }
    if (h->P.env_kind == BP_ENV_BOX) {
        const int E = h->num_envs;
        if (physics) {
            if (mode == MODE_STEP) {
                if (h->steps_done) {
                    hipLaunchKernelGGL(k_make_order, dim3(1), dim3(1024), 0, st, (const unsigned *)h->D.e_cost, h->order_buf, E, (unsigned *)nullptr);
                    HIPCHK(h, hipGetLastError());
                    h->D.order = h->order_buf;
                }
                h->steps_done = true;
                hipLaunchKernelGGL(k_bd_plan, dim3(E), dim3(64), h->bd_lds, st, h->P, h->D, h->B, h->Q, actions);
                HIPCHK(h, hipGetLastError());
                if (!h->st_aux) {
                    HIPCHK(h, hipStreamCreateWithFlags(&h->st_aux, hipStreamNonBlocking));
                    HIPCHK(h, hipStreamCreateWithFlags(&h->st_aux2, hipStreamNonBlocking));
                    HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
                    HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
                    HIPCHK(h, hipEventCreateWithFlags(&h->ev_join2, hipEventDisableTiming));
                    HIPCHK(h, hipStreamCreateWithFlags(&h->st_aux3, hipStreamNonBlocking));
                    HIPCHK(h, hipEventCreateWithFlags(&h->ev_join3, hipEventDisableTiming));
                }
                auto finish = [&](const BdParams &Bx, hipStream_t s_) {
                    if (h->B.task == 1) hipLaunchKernelGGL(k_ac_finish, dim3(E), dim3(64), h->bd_lds, s_, h->P, h->D, Bx, h->Q, 0, 0, reward, term, trunc, info);
                    else hipLaunchKernelGGL(k_bd_finish, dim3(E), dim3(64), h->bd_lds, s_, h->P, h->D, Bx, h->Q, 0, 0, reward, term, trunc, info);
                };
                if (h->bd_budget > 0) {
                    // Two-pass step: a launch of k_bd_physics lasts as long as its slowest env, and the slowest are single wavefronts inside the reference's own
                    // 10 001-step loops.  Pass 0 gives every env `bd_budget` sim steps; the envs that are done -- all but a handful -- go through
                    // finish / robot map / observation on the caller's stream while pass 1 carries the others to their end on a third stream, followed by the
                    // same kernels for that group.  Results are those of the single pass (the loop state travels through Q.rs_*, bodies and arbiters through the
                    // ordinary store / load of a step boundary).
                    BdParams B0 = h->B, B1 = h->B, Bs0 = h->B, Bs1 = h->B;
                    B0.budget = h->bd_budget; B0.pass = 0; B1.pass = 1; Bs0.sel_want = 0; Bs1.sel_want = 1;
                    if (h->damp) hipLaunchKernelGGL(k_bd_physics_damp, dim3(E), dim3(64), h->lds_bytes, st, h->P, h->D, B0, h->Q);
                    else hipLaunchKernelGGL(k_bd_physics, dim3(E), dim3(64), h->lds_bytes, st, h->P, h->D, B0, h->Q);
                    HIPCHK(h, hipGetLastError());
                    HIPCHK(h, hipEventRecord(h->ev_fork, st));
                    // group 1 (unfinished): third stream
                    HIPCHK(h, hipStreamWaitEvent(h->st_aux2, h->ev_fork, 0));
                    if (h->damp) hipLaunchKernelGGL(k_bd_physics_damp, dim3(E), dim3(64), h->lds_bytes, h->st_aux2, h->P, h->D, B1, h->Q);
                    else if (h->B.cycle_skip > 0) {
                        // the envs that stopped inside execute_robot_path: the kernel with the recurrence test (a robot pushing against a wall is over after a
                        // handful of sim steps there); then the others -- the reference's own until-still loops -- on the lean kernel
                        // (side by side on two streams: an env that pushes boxes through a long path is neither)
                        BdParams B1p = B1, B1s = B1;
                        B1p.resume_sel = 1; B1s.resume_sel = 2;
                        HIPCHK(h, hipStreamWaitEvent(h->st_aux3, h->ev_fork, 0));
                        hipLaunchKernelGGL(k_bd_physics_resume, dim3(E), dim3(64), h->lds_bytes, h->st_aux3, h->P, h->D, B1p, h->Q);
                        HIPCHK(h, hipGetLastError());
                        HIPCHK(h, hipEventRecord(h->ev_join3, h->st_aux3));
                        hipLaunchKernelGGL(k_bd_physics, dim3(E), dim3(64), h->lds_bytes, h->st_aux2, h->P, h->D, B1s, h->Q);
                        HIPCHK(h, hipGetLastError());
                        HIPCHK(h, hipStreamWaitEvent(h->st_aux2, h->ev_join3, 0));
                    } else hipLaunchKernelGGL(k_bd_physics, dim3(E), dim3(64), h->lds_bytes, h->st_aux2, h->P, h->D, B1, h->Q);
                    HIPCHK(h, hipGetLastError());
                    hipLaunchKernelGGL(k_bd_robot_map, dim3(E), dim3(BDR_THREADS), h->bd_rmap_lds, h->st_aux2, h->P, h->D, Bs1, h->Q, 0);
                    finish(Bs1, h->st_aux2);
                    HIPCHK(h, hipGetLastError());
                    if (raster && obs) {
                        hipLaunchKernelGGL(k_bd_observe, dim3(E), dim3(BDO_THREADS), h->bd_obs_lds, h->st_aux2, h->P, h->D, Bs1, h->Q, (const unsigned char *)nullptr, obs);
                        HIPCHK(h, hipGetLastError());
                    }
                    HIPCHK(h, hipEventRecord(h->ev_join2, h->st_aux2));
                    // group 0 (done in pass 0): the caller's stream, the robot map beside the finish kernel as before
                    HIPCHK(h, hipStreamWaitEvent(h->st_aux, h->ev_fork, 0));
                    hipLaunchKernelGGL(k_bd_robot_map, dim3(E), dim3(BDR_THREADS), h->bd_rmap_lds, h->st_aux, h->P, h->D, Bs0, h->Q, 0);
                    HIPCHK(h, hipGetLastError());
                    HIPCHK(h, hipEventRecord(h->ev_join, h->st_aux));
                    finish(Bs0, st);
                    HIPCHK(h, hipGetLastError());
                    HIPCHK(h, hipStreamWaitEvent(st, h->ev_join, 0));
                    if (raster && obs) {
                        hipLaunchKernelGGL(k_bd_observe, dim3(E), dim3(BDO_THREADS), h->bd_obs_lds, st, h->P, h->D, Bs0, h->Q, (const unsigned char *)nullptr, obs);
                        HIPCHK(h, hipGetLastError());
                        raster = false;   // both groups' observations are on their way
                    }
                    HIPCHK(h, hipStreamWaitEvent(st, h->ev_join2, 0));
                } else {
                if (h->damp) hipLaunchKernelGGL(k_bd_physics_damp, dim3(E), dim3(64), h->lds_bytes, st, h->P, h->D, h->B, h->Q);
                else hipLaunchKernelGGL(k_bd_physics, dim3(E), dim3(64), h->lds_bytes, st, h->P, h->D, h->B, h->Q);
                HIPCHK(h, hipGetLastError());
                // the robot's spfa map needs only the robot pose: it runs beside the finish kernel on a second stream
                HIPCHK(h, hipEventRecord(h->ev_fork, st));
                HIPCHK(h, hipStreamWaitEvent(h->st_aux, h->ev_fork, 0));
                hipLaunchKernelGGL(k_bd_robot_map, dim3(E), dim3(BDR_THREADS), h->bd_rmap_lds, h->st_aux, h->P, h->D, h->B, h->Q, 0);
                HIPCHK(h, hipGetLastError());
                HIPCHK(h, hipEventRecord(h->ev_join, h->st_aux));
                finish(h->B, st);
                HIPCHK(h, hipGetLastError());
                HIPCHK(h, hipStreamWaitEvent(st, h->ev_join, 0));
                }
            } else {
                hipLaunchKernelGGL(k_reset_copy, dim3(E), dim3(256), 0, st, h->P, h->D, mask, (double *)nullptr);
                HIPCHK(h, hipGetLastError());
                hipLaunchKernelGGL(k_bd_reset_copy, dim3(E), dim3(256), 0, st, h->P, h->D, h->B, h->Q, mask, info);
                HIPCHK(h, hipGetLastError());
            }
        }
        if (h->timing) HIPCHK(h, hipEventRecord(e1, st));
        if (raster && obs) {
            hipLaunchKernelGGL(k_bd_observe, dim3(E), dim3(BDO_THREADS), h->bd_obs_lds, st, h->P, h->D, h->B, h->Q, mask, obs);
            HIPCHK(h, hipGetLastError());
        }
        if (h->timing) HIPCHK(h, hipEventRecord(e2, st));
        return BP_OK;
    }
    if (physics) {
        if (mode == MODE_STEP && h->steps_done) { // heaviest-first dispatch order from the previous step's per-env cycles
            hipLaunchKernelGGL(k_make_order, dim3(1), dim3(1024), 0, st, (const unsigned *)h->D.e_cost, h->order_buf, h->num_envs, h->D.sq_thr);
            HIPCHK(h, hipGetLastError());
            h->D.order = h->order_buf;
        }
        if (mode == MODE_STEP) h->steps_done = true;
        if (mode == MODE_STEP && h->pair_mode == 1 && h->D.dbg == nullptr)
            hipLaunchKernelGGL(k_physics_step_pair, dim3((h->num_envs + 1) / 2), dim3(64), 2 * PL_HALF, st, h->P, h->D, actions, reward, term, trunc, info);
        else if (mode == MODE_STEP && h->sched_chunk > 0 && h->D.dbg == nullptr) {
            // preemptive scheduler: one workgroup per (env, chunk) task (most leave at once: only parked envs need a second workgroup)
            hipLaunchKernelGGL(k_sched_init, dim3(16), dim3(1024), 0, st, h->P, h->D);
            HIPCHK(h, hipGetLastError());
            // resident launches only while this process has the device to itself (see resident_device_shared)
            if (h->resident_auto && (++h->launches & 63u) == 0) h->device_shared = resident_device_shared(h);
            const bool resident_ok = !(h->resident_auto && h->device_shared);
            if (h->sched_persist && resident_ok) {
                // resident wavefronts: one workgroup per wave slot for the whole launch; the launch constants go through device memory (see sched_resident)
                DevParams *Pg = (DevParams *)h->pd_buf;
                DevPtrs *Dg = (DevPtrs *)((char *)h->pd_buf + sizeof(DevParams));
                hipLaunchKernelGGL(k_store_params, dim3(1), dim3(64), 0, st, h->P, h->D, Pg, Dg);
                HIPCHK(h, hipGetLastError());
                if (h->maze8)
                    hipLaunchKernelGGL(k_physics_step_schedl_maze, dim3(h->sched_persist), dim3(64), h->sched_lds, st, Pg, Dg, actions, reward, term, trunc, info);
                else
                    hipLaunchKernelGGL(k_physics_step_schedl, dim3(h->sched_persist), dim3(64), h->sched_lds, st, Pg, Dg, actions, reward, term, trunc, info);
            } else if (h->pair_resident > 0 && h->P.pair_mode == 2 && resident_ok) {
                // a pairing launch on resident wavefronts: one kernel, the two step bodies as functions of their own (k_physics_step_schedr)
                DevParams *Pg = (DevParams *)h->pd_buf;
                DevPtrs *Dg = (DevPtrs *)((char *)h->pd_buf + sizeof(DevParams));
                hipLaunchKernelGGL(k_store_params, dim3(1), dim3(64), 0, st, h->P, h->D, Pg, Dg);
                HIPCHK(h, hipGetLastError());
                hipLaunchKernelGGL(k_physics_step_schedr, dim3(h->pair_resident), dim3(64), h->sched_lds, st, Pg, Dg, actions, reward, term, trunc, info);
            } else if (h->maze8)
                hipLaunchKernelGGL(k_physics_step_sched_maze, dim3(h->num_envs * h->P.sq_levels), dim3(64), h->sched_lds, st, h->P, h->D, actions, reward, term, trunc, info);
            else if (h->P.pair_mode == 2) {
                // (BP_PAIR_RESIDENT=0) a pairing launch as two kernels side by side: the envs that start alone on the lean solo code (second stream), everything else -- paired
                // first tasks, the workgroups that serve the queues -- in the kernel that holds both step bodies
                if (!h->st_aux) {
                    HIPCHK(h, hipStreamCreateWithFlags(&h->st_aux, hipStreamNonBlocking));
                    HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
                    HIPCHK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
                }
                if (h->P.pair_solo > 0) {
                    HIPCHK(h, hipEventRecord(h->ev_fork, st));
                    HIPCHK(h, hipStreamWaitEvent(h->st_aux, h->ev_fork, 0));
                    hipLaunchKernelGGL(k_physics_step_sched, dim3(h->P.pair_solo), dim3(64), h->sched_lds, h->st_aux, h->P, h->D, actions, reward, term, trunc, info);
                    HIPCHK(h, hipGetLastError());
                    HIPCHK(h, hipEventRecord(h->ev_join, h->st_aux));
                }
                hipLaunchKernelGGL(k_physics_step_schedp, dim3(h->num_envs * h->P.sq_levels), dim3(64), h->sched_lds, st, h->P, h->D, actions, reward, term, trunc, info);
                HIPCHK(h, hipGetLastError());
                if (h->P.pair_solo > 0) HIPCHK(h, hipStreamWaitEvent(st, h->ev_join, 0));
            } else if (h->P.sq_parts > 1) {
                // the scheduled launch as several kernels on several streams (hardware queues)
                const int np = h->P.sq_parts;
                if (h->st_parts.empty()) {
                    for (int k = 0; k < 7; k++) { hipStream_t s_; HIPCHK(h, hipStreamCreateWithFlags(&s_, hipStreamNonBlocking)); h->st_parts.push_back(s_); }
                    for (int k = 0; k < 8; k++) { hipEvent_t e_; HIPCHK(h, hipEventCreateWithFlags(&e_, hipEventDisableTiming)); h->ev_parts.push_back(e_); }
                }
                HIPCHK(h, hipEventRecord(h->ev_parts[7], st));
                for (int k = 0; k < np; k++) {
                    DevParams Pk = h->P;
                    Pk.sq_part = k;
                    hipStream_t sk = (k == 0) ? st : h->st_parts[k - 1];
                    if (k > 0) HIPCHK(h, hipStreamWaitEvent(sk, h->ev_parts[7], 0));
                    const int grid = (h->num_envs * h->P.sq_levels + np - 1) / np;
                    hipLaunchKernelGGL(k_physics_step_sched, dim3(grid), dim3(64), h->sched_lds, sk, Pk, h->D, actions, reward, term, trunc, info);
                    HIPCHK(h, hipGetLastError());
                    if (k > 0) { HIPCHK(h, hipEventRecord(h->ev_parts[k - 1], sk)); }
                }
                for (int k = 1; k < np; k++) HIPCHK(h, hipStreamWaitEvent(st, h->ev_parts[k - 1], 0));
            } else
                hipLaunchKernelGGL(k_physics_step_sched, dim3(h->num_envs * h->P.sq_levels), dim3(64), h->sched_lds, st, h->P, h->D, actions, reward, term, trunc, info);
            HIPCHK(h, hipGetLastError());
            // completion launch: workgroup b finishes the b-th env that the scheduled launch left unfinished (scheduler watchdog); normally all leave at once
#ifdef BP_DEBUG_PATHS
            static const bool completion = !(getenv("BP_SCHED_COMPLETION") && atoi(getenv("BP_SCHED_COMPLETION")) == 0);   // diagnostic switch (twin library only)
#else
            constexpr bool completion = true;
#endif
            if (completion) {
                hipLaunchKernelGGL(k_sched_scan, dim3(1), dim3(256), 0, st, h->P, h->D);
                HIPCHK(h, hipGetLastError());
                DevParams PC = h->P;
                PC.sq_mode = 1;
                if (h->maze8)
                    hipLaunchKernelGGL(k_physics_step_sched_maze, dim3(std::min(h->num_envs, SQ_RESCUE)), dim3(64), h->sched_lds, st, PC, h->D, actions, reward, term, trunc, info);
                else
                    hipLaunchKernelGGL(k_physics_step_sched, dim3(std::min(h->num_envs, SQ_RESCUE)), dim3(64), h->sched_lds, st, PC, h->D, actions, reward, term, trunc, info);
            }
        }
        else if (mode == MODE_STEP && h->damp)
            hipLaunchKernelGGL(k_physics_step_damp, dim3(h->num_envs), dim3(64), h->lds_bytes, st, h->P, h->D, actions, reward, term, trunc, info);
        else if (h->resettle && h->damp)
            hipLaunchKernelGGL(k_physics_reset_damp, dim3(h->num_envs), dim3(64), h->lds_bytes, st, h->P, h->D, mask, info, 0);
        else if (mode == MODE_STEP && h->maze8)
            hipLaunchKernelGGL(k_physics_step_maze, dim3(h->num_envs), dim3(64), h->lds_bytes, st, h->P, h->D, actions, reward, term, trunc, info);
        else if (mode == MODE_STEP)
            hipLaunchKernelGGL(k_physics_step, dim3(h->num_envs), dim3(64), h->lds_bytes, st, h->P, h->D, actions, reward, term, trunc, info);
        else if (h->resettle && h->maze8)
            hipLaunchKernelGGL(k_physics_reset_maze, dim3(h->num_envs), dim3(64), h->lds_bytes, st, h->P, h->D, mask, info, 0);
        else if (h->resettle)
            hipLaunchKernelGGL(k_physics_reset, dim3(h->num_envs), dim3(64), h->lds_bytes, st, h->P, h->D, mask, info, 0);
        else
            hipLaunchKernelGGL(k_reset_copy, dim3(h->num_envs), dim3(256), 0, st, h->P, h->D, mask, info);
        HIPCHK(h, hipGetLastError());
        if (mode == MODE_STEP && h->timing && h->cost_ring && h->cost_n < BP_COST_RING) {
            hipLaunchKernelGGL(k_cost_stats, dim3(1), dim3(1024), 0, st, (const unsigned *)h->D.e_cost, h->num_envs, h->cost_ring, h->cost_n);
            HIPCHK(h, hipGetLastError());
            h->cost_n++;
        }
        if (h->P.env_kind == BP_ENV_SHIP_ICE || h->P.env_kind == BP_ENV_MAZE) {
            hipLaunchKernelGGL(k_episode_metrics, dim3((h->num_envs + 255) / 256), dim3(256), 0, st, h->P, h->D, mode == MODE_STEP ? 0 : 1, mask);
            HIPCHK(h, hipGetLastError());
        }
    }
    if (h->timing) HIPCHK(h, hipEventRecord(e1, st));
    if (raster && obs) {
        if (h->P.env_kind == BP_ENV_MAZE)
            hipLaunchKernelGGL(k_observe_maze, dim3(h->num_envs), dim3(OBS_THREADS), h->obs_lds_bytes, st, h->P, h->D, mask, obs);
        else
            hipLaunchKernelGGL(k_observe, dim3(h->num_envs), dim3(OBS_THREADS_SHIP), h->obs_lds_bytes, st, h->P, h->D, mask, obs);
        HIPCHK(h, hipGetLastError());
    }
    if (h->timing) HIPCHK(h, hipEventRecord(e2, st));
    return BP_OK;
}

int bp_reset(bp_handle *h, const uint8_t *env_mask, uint8_t *obs, double *info, void *stream)
{
    if (!h) return BP_EINVAL;
    if (!h->loaded) return fail(h, BP_ESTATE, "bp_load_scenarios has not been called");
    BP_DEVICE(h);
    const bool save = h->timing;
    h->timing = false; // resets are not part of the per-step kernel timing
    const int rc = launch(h, MODE_RESET, nullptr, env_mask, obs, nullptr, nullptr, nullptr, info, (hipStream_t)stream, true, true);
    h->timing = save;
    if (rc == BP_OK) h->was_reset = true;
    return rc;
}

int bp_step(bp_handle *h, const double *actions, uint8_t *obs, double *reward, uint8_t *terminated, uint8_t *truncated,
            double *info, void *stream)
{
    if (!h || !actions) return BP_EINVAL;
    if (!h->loaded || !h->was_reset) return fail(h, BP_ESTATE, "bp_step before bp_load_scenarios/bp_reset");
    BP_DEVICE(h);
    return launch(h, MODE_STEP, actions, nullptr, obs, reward, terminated, truncated, info, (hipStream_t)stream, true, true);
}

int bp_step_physics(bp_handle *h, const double *actions, double *reward, uint8_t *terminated, uint8_t *truncated, double *info,
                    void *stream)
{
    if (!h || !actions) return BP_EINVAL;
    if (!h->loaded || !h->was_reset) return fail(h, BP_ESTATE, "bp_step_physics before bp_load_scenarios/bp_reset");
    BP_DEVICE(h);
    return launch(h, MODE_STEP, actions, nullptr, nullptr, reward, terminated, truncated, info, (hipStream_t)stream, true, false);
}

int bp_observe(bp_handle *h, const uint8_t *env_mask, uint8_t *obs, void *stream)
{
    if (!h || !obs) return BP_EINVAL;
    if (!h->loaded || !h->was_reset) return fail(h, BP_ESTATE, "bp_observe before bp_load_scenarios/bp_reset");
    BP_DEVICE(h);
    return launch(h, MODE_STEP, nullptr, env_mask, obs, nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream, false, true);
}

int bp_observe_global(bp_handle *h, const uint8_t *env_mask, uint8_t *obs, void *stream)
{
    if (!h || !obs) return BP_EINVAL;
    if (!h->loaded || !h->was_reset) return fail(h, BP_ESTATE, "bp_observe_global before bp_load_scenarios/bp_reset");
    if (h->P.env_kind != BP_ENV_SHIP_ICE) return fail(h, BP_ESTATE, "global observation exists for ship-ice only");
    BP_DEVICE(h);
    const int cell_px = (int)(0.2 * h->cfg.m_to_pix); // grid_width 0.2 m (ship_ice_env.py:97), block = int(0.2 * 25) = 5 px
    if (cell_px <= 0 || h->P.grid_h % cell_px || h->P.grid_w % cell_px) return fail(h, BP_EINVAL, "raster is not a multiple of the cell");
    const size_t lds = (size_t)4 * ((h->P.grid_h * h->P.grid_w + 31) / 32);
    HIPCHK(h, hipFuncSetAttribute((const void *)k_observe_global, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_observe_global, dim3(h->num_envs), dim3(OBS_THREADS), lds, (hipStream_t)stream, h->P, h->D, env_mask, obs, cell_px);
    HIPCHK(h, hipGetLastError());
    return BP_OK;
}

__global__ void k_export_polys(const DevParams P, const DevPtrs D, double *out, int *counts)
{
    const int env = blockIdx.x;
    const size_t eb = (size_t)env * P.nbcap;
    const int nb = D.e_nb[env];
    const int *nv = D.sc_nv + (size_t)D.e_trial[env] * P.nbcap;
    for (int i = threadIdx.x; i < P.nbcap; i += blockDim.x) {
        const int n = (i < nb) ? nv[i] : 0;
        counts[eb + i] = n;
        for (int q = 0; q < BP_MAXV; q++) {
            const d2 v = (q < n) ? D.wv[(eb + i) * BP_MAXV + q] : mk2(0.0, 0.0);
            out[((eb + i) * BP_MAXV + q) * 2] = v.x;
            out[((eb + i) * BP_MAXV + q) * 2 + 1] = v.y;
        }
    }
}
__global__ void k_export_bodies(const DevParams P, const DevPtrs D, double *out)
{
    const int env = blockIdx.x;
    const size_t eb = (size_t)env * P.nbcap;
    const int nb = D.e_nb[env];
    for (int i = threadIdx.x; i < P.nbcap; i += blockDim.x) {
        double *o = out + (eb + i) * 9;
        if (i < nb) {
            o[0] = D.pxy[eb + i].x; o[1] = D.pxy[eb + i].y; o[2] = D.ang[eb + i];
            o[3] = D.velv[eb + i].x; o[4] = D.velv[eb + i].y; o[5] = D.velw[eb + i].x;
            o[6] = D.velb[eb + i].x; o[7] = D.velb[eb + i].y; o[8] = D.velw[eb + i].y;
        } else {
            for (int k = 0; k < 9; k++) o[k] = 0.0;
        }
    }
}
// generate_observation_low_dim (ship_ice_env.py:358-370): |centroid| of each floe's world polygon
__global__ void k_export_lowdim(const DevParams P, const DevPtrs D, double *out)
{
    const int env = blockIdx.x;
    const size_t eb = (size_t)env * P.nbcap;
    const int nb = D.e_nb[env];
    const int *nv = D.sc_nv + (size_t)D.e_trial[env] * P.nbcap;
    for (int i = 1 + threadIdx.x; i < P.nbcap; i += blockDim.x) {
        double *o = out + ((size_t)env * (P.nbcap - 1) + (i - 1)) * 2;
        if (i < nb) {
            const d2 c = poly_centroid_seq(D.wv + (eb + i) * BP_MAXV, nv[i]);
            o[0] = __builtin_fabs(c.x); o[1] = __builtin_fabs(c.y);
        } else { o[0] = 0.0; o[1] = 0.0; }
    }
}

int bp_get_world_polys(bp_handle *h, double *out, int32_t *counts, void *stream)
{
    if (!h || !out || !counts) return BP_EINVAL;
    if (!h->loaded || !h->was_reset) return fail(h, BP_ESTATE, "not reset");
    BP_DEVICE(h);
    hipLaunchKernelGGL(k_export_polys, dim3(h->num_envs), dim3(256), 0, (hipStream_t)stream, h->P, h->D, out, counts);
    HIPCHK(h, hipGetLastError());
    return BP_OK;
}
double bp_start_uniform(uint64_t seed, int64_t global_env_id, int64_t episode) { return bp_start_u01(seed, global_env_id, episode); }

int bp_get_episode_metrics(bp_handle *h, double *rows, uint32_t *counts, void *stream)
{
    if (!h) return BP_EINVAL;
    if (!h->loaded) return fail(h, BP_ESTATE, "not loaded");
    if (h->P.env_kind != BP_ENV_SHIP_ICE && h->P.env_kind != BP_ENV_MAZE)
        return fail(h, BP_EINVAL, "episode metrics are kept for ship-ice and maze handles only");
    BP_DEVICE(h);
    if (rows) HIPCHK(h, hipMemcpyAsync(rows, h->D.m_rows, sizeof(double) * BP_EPM_COUNT * h->num_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (counts) HIPCHK(h, hipMemcpyAsync(counts, h->D.m_count, sizeof(unsigned) * h->num_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BP_OK;
}

int bp_get_episode_history(bp_handle *h, double *ring, double *sums, uint32_t *counts, void *stream)
{
    if (!h) return BP_EINVAL;
    if (!h->loaded) return fail(h, BP_ESTATE, "not loaded");
    if (h->P.env_kind != BP_ENV_SHIP_ICE && h->P.env_kind != BP_ENV_MAZE)
        return fail(h, BP_EINVAL, "episode metrics are kept for ship-ice and maze handles only");
    BP_DEVICE(h);
    const size_t E = (size_t)h->num_envs;
    if (ring) HIPCHK(h, hipMemcpyAsync(ring, h->D.m_ring, sizeof(double) * BP_EPM_RING * BP_EPM_COUNT * E, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (sums) HIPCHK(h, hipMemcpyAsync(sums, h->D.m_sum, sizeof(double) * BP_EPM_COUNT * E, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (counts) HIPCHK(h, hipMemcpyAsync(counts, h->D.m_count, sizeof(unsigned) * E, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BP_OK;
}

// test hook: every written hint word of the live envs gets random VALID contents -- plane and support-vertex indices below the vertex counts the word
// records, a random subset of the HW_* flags.  The sub-step evaluates cached planes exactly and searches whatever they do not certify, so the results
// of the following steps must not change (tests/test_gpu_parity.py); only the amount of work does.
__global__ __launch_bounds__(256) void k_debug_scramble_hints(unsigned long long *hint, size_t n, unsigned long long seed, int *count)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long hw = hint[i];
    const unsigned nA = (unsigned)HW_NV_A(hw), nB = (unsigned)HW_NV_B(hw);
    if (nA == 0 || nB == 0) return; // never written (or cleared by a neighbour-list refresh)
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1); // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    const unsigned iA = (unsigned)(z & 0xFF) % nA, iB = (unsigned)((z >> 8) & 0xFF) % nB, jA = (unsigned)((z >> 16) & 0xFF) % nB, jB = (unsigned)((z >> 24) & 0xFF) % nA;
    hint[i] = (unsigned long long)(iA | (iB << 5) | (jA << 10) | (jB << 15) | (nA << 20) | (nB << 25)) |
              (((z >> 32) & 1ull) ? HW_HAS_A : 0ull) | (((z >> 33) & 1ull) ? HW_HAS_B : 0ull) | (((z >> 34) & 1ull) ? HW_PRIM_B : 0ull) |
              (((z >> 35) & 1ull) ? HW_BOTH : 0ull);
    atomicAdd(count, 1);
}

int bp_debug_scramble_hints(bp_handle *h, uint64_t seed, void *stream)
{
    if (!h) return BP_EINVAL;
    if (!h->loaded || !h->was_reset) return fail(h, BP_ESTATE, "not reset");
    DevGuard _dg(h->device);
    const size_t n = (size_t)h->num_envs * (size_t)h->nbcap * BP_KADJ;
    int *cnt = nullptr, host = 0;
    HIPCHK(h, hipMalloc(&cnt, sizeof(int)));
    hipError_t e = hipMemsetAsync(cnt, 0, sizeof(int), (hipStream_t)stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_scramble_hints, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->D.hint, n, (unsigned long long)seed, cnt);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&host, cnt, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(cnt);   // also on the error paths
    HIPCHK(h, e);
    return host; // number of hint words rewritten (>= 0; negative = BP_E*)
}

// rows of `src` whose mask byte is set -> the same rows of `dst`; one workgroup per row, 16-byte accesses when the row allows them
__global__ __launch_bounds__(256) void k_copy_rows_masked(const unsigned char *__restrict__ mask, const unsigned *__restrict__ src, unsigned *__restrict__ dst,
                                                          const long long row_words)
{
    const long long r = blockIdx.x;
    if (mask[r] == 0) return;
    const unsigned *s = src + r * row_words;
    unsigned *d = dst + r * row_words;
    if ((row_words & 3) == 0 && ((((size_t)s) | ((size_t)d)) & 15) == 0) {
        const uint4 *s4 = (const uint4 *)s; uint4 *d4 = (uint4 *)d;
        for (long long i = threadIdx.x; i < row_words / 4; i += blockDim.x) d4[i] = s4[i];
    } else
        for (long long i = threadIdx.x; i < row_words; i += blockDim.x) d[i] = s[i];
}

int bp_copy_rows_masked(bp_handle *h, const uint8_t *mask, const void *src, void *dst, int64_t rows, int64_t row_bytes, void *stream)
{
    if (!h || !mask || !src || !dst || rows < 0 || row_bytes <= 0 || (row_bytes & 3) != 0 || rows > 0x7FFFFFFF) return BP_EINVAL;
    if ((((size_t)src) | ((size_t)dst)) & 3) return BP_EINVAL;
    BP_DEVICE(h);
    if (rows == 0) return BP_OK;
    hipLaunchKernelGGL(k_copy_rows_masked, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const unsigned char *)mask, (const unsigned *)src,
                       (unsigned *)dst, (long long)(row_bytes / 4));
    HIPCHK(h, hipGetLastError());
    return BP_OK;
}

int bp_debug_round2(const double *in_dev, double *out_dev, int32_t n, void *stream)
{
    if (!in_dev || !out_dev || n < 0) return BP_EINVAL;
    hipLaunchKernelGGL(k_debug_round2, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev, n);
    return hipGetLastError() == hipSuccess ? BP_OK : BP_EHIP;
}

int bp_get_body_state(bp_handle *h, double *out, void *stream)
{
    if (!h || !out) return BP_EINVAL;
    if (!h->loaded || !h->was_reset) return fail(h, BP_ESTATE, "not reset");
    BP_DEVICE(h);
    hipLaunchKernelGGL(k_export_bodies, dim3(h->num_envs), dim3(256), 0, (hipStream_t)stream, h->P, h->D, out);
    HIPCHK(h, hipGetLastError());
    return BP_OK;
}
int bp_costmap_update(bp_handle *h, const bp_costmap_config *cfg, const double *ship_pos_y, double vs, double *out, void *stream)
{
    if (!h || !cfg || !out) return BP_EINVAL;
    if (!h->loaded || !h->was_reset) return fail(h, BP_ESTATE, "bp_costmap_update before bp_load_scenarios/bp_reset");
    if (h->P.env_kind != BP_ENV_SHIP_ICE) return fail(h, BP_EINVAL, "bp_costmap_update: ship-ice handles only");
    const int H = (int)(cfg->m * cfg->scale), W = (int)(cfg->n * cfg->scale);
    if (!(cfg->scale > 0) || H <= 0 || W <= 0 || cfg->margin < 0 || 2 * cfg->margin > W) return fail(h, BP_EINVAL, "bp_costmap_update: bad grid");
    BP_DEVICE(h);
    hipStream_t st = (hipStream_t)stream;
    const size_t total = (size_t)h->num_envs * H * W;
    const int nblk = (int)std::min<size_t>((total + 255) / 256, 65535);
    hipLaunchKernelGGL(k_costmap_init, dim3(nblk), dim3(256), 0, st, out, h->num_envs, H, W, (int)cfg->margin);
    const int groups = (h->P.nbcap - h->P.nkin + 3) / 4;
    if (groups > 0)
        hipLaunchKernelGGL(k_costmap, dim3(h->num_envs, groups), dim3(256), 0, st, h->P, h->D, cfg->scale, H, W, cfg->alpha, cfg->ship_mass,
                           cfg->horizon > 0 ? cfg->horizon * cfg->scale : 0.0, ship_pos_y, vs, out);
    HIPCHK(h, hipGetLastError());
    return BP_OK;
}
int bp_get_low_dim_obs(bp_handle *h, double *out, void *stream)
{
    if (!h || !out) return BP_EINVAL;
    if (!h->loaded || !h->was_reset) return fail(h, BP_ESTATE, "not reset");
    BP_DEVICE(h);
    hipLaunchKernelGGL(k_export_lowdim, dim3(h->num_envs), dim3(256), 0, (hipStream_t)stream, h->P, h->D, out);
    HIPCHK(h, hipGetLastError());
    return BP_OK;
}

int32_t bp_nb_cap(const bp_handle *h) { return h ? h->nbcap : 0; }
int32_t bp_obs_height(const bp_handle *h) { return h ? h->P.obs_h : 0; }
int32_t bp_obs_width(const bp_handle *h) { return h ? h->P.obs_w : 0; }

// ---- box-delivery-v0 ---------------------------------------------------------------------------------------------------
int32_t bp_bd_sizeof_config(void) { return (int32_t)sizeof(bp_bd_config); }

int bp_bd_create(const bp_bd_config *cfg, int32_t num_envs, int64_t env_id_offset, int32_t device, bp_handle **out)
{
    if (!cfg || !out || num_envs <= 0) return BP_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BP_ENODEVICE;
    if (!(cfg->damping_pow >= 0.0) || cfg->damping_pow > 1.0 || cfg->steps <= 0 || cfg->iterations <= 0 || cfg->persistence <= 0) return BP_EINVAL;
    if (cfg->num_boxes <= 0 || cfg->num_boxes > BD_MAXBOX || cfg->local_px <= 0 || (cfg->local_px & 1)) return BP_EINVAL;
    bp_handle *h = new bp_handle();
    memset(&h->cfg, 0, sizeof(h->cfg));
    h->bdcfg = *cfg;
    h->num_envs = num_envs; h->env_offset = env_id_offset; h->device = device;
    DevGuard _dg(device);
    if (!_dg.ok) { delete h; return BP_ENODEVICE; }
    memset(&h->D, 0, sizeof(h->D));
    memset(&h->Q, 0, sizeof(h->Q));
    DevParams &P = h->P;
    memset(&P, 0, sizeof(P));
    P.dt_sub = cfg->ctrl_dt / cfg->steps;
    P.steps = cfg->steps; P.iterations = cfg->iterations; P.persistence = cfg->persistence; P.settle_steps = cfg->settle_steps;
    if (const char *evp = getenv("BP_DEBUG_PATHS")) P.dbg_paths = atoi(evp); // test hook: force the rarely taken paths of the narrow phase (bp_device.hpp)
    P.damping_pow = cfg->damping_pow; P.bias_coef = cfg->bias_coef; P.slop = cfg->slop;
    P.target_speed = cfg->target_speed;
    P.skin = 0.25;
    P.env_kind = BP_ENV_BOX;
    P.nkin = 6;                                            // main shape + 4 wheels + front bumper on one KINEMATIC body
    P.num_envs = num_envs; P.env_offset = env_id_offset;
    P.obs_h = P.obs_w = cfg->local_px;
    BdParams &B = h->B;
    memset(&B, 0, sizeof(B));
    B.room_length = cfg->room_length; B.room_width = cfg->room_width; B.recept_x = cfg->recept_x; B.recept_y = cfg->recept_y;
    B.recept_size = cfg->recept_size; B.ppm = cfg->ppm; B.local_w = cfg->local_w; B.robot_radius = cfg->robot_radius;
    B.step_size = cfg->step_size; B.target_speed = cfg->target_speed; B.ctrl_dt = cfg->ctrl_dt;
    B.partial_rewards_scale = cfg->partial_rewards_scale; B.goal_reward = cfg->goal_reward; B.collision_penalty = cfg->collision_penalty;
    B.non_movement_penalty = cfg->non_movement_penalty; B.correct_direction_reward_scale = cfg->correct_direction_reward_scale;
    B.ministep_size = cfg->ministep_size; B.sp_channel_scale = cfg->sp_channel_scale;
    B.local_px = cfg->local_px; B.use_correct_direction_reward = cfg->use_correct_direction_reward;
    B.inactivity_cutoff = cfg->inactivity_cutoff; B.num_boxes = cfg->num_boxes; B.step_limit = cfg->step_limit;
    B.first_box = 6;
    B.action_type = cfg->action_type;
    if (cfg->action_type < 0 || cfg->action_type > 2) { delete h; return BP_EINVAL; }
    B.task = cfg->task; B.omega_scale = cfg->omega_scale; B.v_scale = cfg->v_scale; B.lfc = cfg->lfc;
    B.yaw_rate_step = cfg->yaw_rate_step; B.t_max = cfg->t_max;
    B.boundary_penalty = cfg->boundary_penalty; B.box_cleared_reward = cfg->box_cleared_reward; B.box_putback_penalty = cfg->box_putback_penalty;
    B.truncation_penalty = cfg->truncation_penalty; B.terminal_reward = cfg->terminal_reward; B.pushing_mult = cfg->pushing_mult;
    // outside the small-map window every cell is padding obstacle: spfa leaves 0 there, the inverted map (box_delivery_env.py:1126-1128) 0 + 1
    B.recept_outside = (cfg->task == 0 && cfg->invert_receptacle_map) ? 1.0f : 0.0f;
    if (cfg->task == 1) {
        if (cfg->num_boundary_verts < 3 || cfg->num_boundary_verts > 8 || cfg->num_outer_verts < 3 || cfg->num_outer_verts > 8 ||
            cfg->num_goal_points < 1 || cfg->num_goal_points > 128) { delete h; return BP_EINVAL; }
        B.nbd = cfg->num_boundary_verts; B.nob = cfg->num_outer_verts; B.ngoal = cfg->num_goal_points;
        memcpy(B.bd_poly, cfg->boundary, sizeof(B.bd_poly)); memcpy(B.ob_poly, cfg->outer_boundary, sizeof(B.ob_poly));
        memcpy(B.footprint, cfg->footprint_verts, sizeof(B.footprint));
        B.recept_outside = 1.0f;
    } else if (cfg->task != 0) { delete h; return BP_EINVAL; }
    *out = h;
    return BP_OK;
}

int bp_bd_load(bp_handle *h, int32_t T, int32_t nbox, const double *starts, const double *boxes, int32_t ns, const double *sverts,
               const int32_t *scount, const double *spose, const double *srad, const int32_t *stype)
{
    if (!h || T <= 0 || nbox <= 0 || nbox > BD_MAXBOX || ns <= 0 || !starts || !boxes || !sverts || !scount || !spose || !srad || !stype) return BP_EINVAL;
    if (h->loaded) return fail(h, BP_ESTATE, "scenarios already loaded");
    if (h->P.env_kind != BP_ENV_BOX) return fail(h, BP_ESTATE, "handle was created for another environment");
    BP_DEVICE(h);
    using namespace bpgeom;
    const bp_bd_config &cf = h->bdcfg;
    std::vector<std::vector<Shape>> trials(T);
    std::vector<Shape> recepts(T);
    h->bd_map_of_trial.assign(T, 0);
    std::vector<std::vector<std::vector<P2>>> map_keys; // obstacle polygons of each distinct layout
    int nphys_static = -1;
    for (int t = 0; t < T; t++) {
        std::vector<Shape> &bodies = trials[t];
        const double sx = starts[3 * t], sy = starts[3 * t + 1], sh = starts[3 * t + 2];
        {   // create_agent (sim_utils.py:20-73): main shape radius 0 / friction 1, wheels + bumper radius 0.02 / friction 0
            Shape s;
            build_agent_main(cf.robot_verts, 4, sx, sy, sh, s);
            s.radius = 0.0; s.e = 0.01; s.u = 1.0; s.kind = kind_of(1, 1, BODY_KINEMATIC);
            bodies.push_back(s);
            for (int k = 0; k < 5; k++) {
                Shape w;
                build_kinematic_part(k < 4 ? cf.wheel_verts[k] : cf.bumper_verts, 4, sx, sy, sh, w);
                w.radius = 0.02; w.e = 0.01; w.u = 0.0; w.kind = kind_of(0, 1, BODY_KINEMATIC);
                bodies.push_back(w);
            }
        }
        for (int b = 0; b < nbox; b++) {
            const double *bx = boxes + ((size_t)t * nbox + b) * 3;
            Shape s;
            if (!build_box(bx[0], bx[1], bx[2], cf.box_half, cf.box_density, 0.02, s)) return fail(h, BP_EINVAL, "degenerate box");
            s.radius = 0.02; s.e = 0.01; s.u = 1.0; s.kind = kind_of(2, 0, BODY_DYNAMIC);
            bodies.push_back(s);
        }
        std::vector<std::vector<P2>> obstacles;
        int nrec = 0, nst = 0;
        for (int k = 0; k < ns; k++) {
            const size_t o = (size_t)t * ns + k;
            const int n = scount[o];
            if (n == 0) continue;   // padding: trials may hold different numbers of columns
            if (n < 3 || n > 4) return fail(h, BP_EINVAL, "static polygons have 3 or 4 vertices");
            Shape s;
            build_static_poly(sverts + o * 8, n, spose[o * 3], spose[o * 3 + 1], spose[o * 3 + 2], s);
            s.radius = srad[o]; s.e = 0.01; s.u = 1.0;
            if (cf.task == 1) { s.e = 0.0; s.u = 0.99; }   // area_clearing.py:446-474: pymunk default elasticity, friction 0.99
            if (stype[o] == 4) {
                if (s.verts.size() != 4) return fail(h, BP_EINVAL, "the receptacle must be a quadrilateral");
                recepts[t] = s; nrec++;
                continue;            // the receptacle never produces a collision response (handlers :216-229): not a physics slot
            }
            s.kind = kind_of(3, 0, BODY_STATIC);
            bodies.push_back(s);
            obstacles.push_back(world_verts(s));
            nst++;
        }
        if (cf.task == 0 && nrec != 1) return fail(h, BP_EINVAL, "exactly one receptacle polygon per trial is required");
        if (cf.task == 1 && nrec != 0) return fail(h, BP_EINVAL, "area-clearing has no receptacle polygon");
        (void)nphys_static; (void)nst;
        int mi = -1;
        for (size_t m = 0; m < map_keys.size() && mi < 0; m++) {
            bool same = map_keys[m].size() == obstacles.size();
            for (size_t q = 0; same && q < obstacles.size(); q++) {
                same = map_keys[m][q].size() == obstacles[q].size();
                for (size_t v = 0; same && v < obstacles[q].size(); v++) same = map_keys[m][q][v].x == obstacles[q][v].x && map_keys[m][q][v].y == obstacles[q][v].y;
            }
            if (same) mi = (int)m;
        }
        if (mi < 0) {
            BdMaps M;
            AcGeom G;
            G.nbd = cf.num_boundary_verts; G.nob = cf.num_outer_verts; G.ngoal = cf.num_goal_points;
            G.bd = cf.boundary; G.ob = cf.outer_boundary; G.goals = cf.goal_points; G.scale_max = cf.distance_scale_max;
            if (!bd_build_maps(obstacles, cf.room_length, cf.room_width, cf.ppm, cf.local_px, cf.local_w, cf.robot_radius, cf.robot_half_width,
                               cf.recept_x, cf.recept_y, cf.sp_channel_scale, M, cf.task, &G, cf.task == 0 && cf.invert_receptacle_map != 0))
                return fail(h, BP_EINVAL, "free space does not fit the small-map window");
            h->bd_maps.push_back(M);
            map_keys.push_back(obstacles);
            mi = (int)map_keys.size() - 1;
        }
        h->bd_map_of_trial[t] = mi;
    }
    int rc = upload_trials(h, trials, false);
    if (rc) return rc;
    if (h->nbcap > 64) return fail(h, BP_EINVAL, "box-delivery supports at most 64 shape slots per env");
    BdParams &B = h->B;
    const BdMaps &M0 = h->bd_maps[0];
    B.H = M0.H; B.W = M0.W; B.SH = M0.SH; B.SW = M0.SW; B.si0 = M0.si0; B.sj0 = M0.sj0;
    B.nbox = nbox; B.nrecept = 1; B.out_r = M0.out_r;
    B.budget = 0; B.pass = 0; B.sel_want = -1; B.resume_sel = 0;
    // execute_robot_path skips whole periods once the robot's state recurs exactly (a robot that pushes against a wall until STEP_LIMIT); BP_BD_CYCLE=0: every sim step is run
    // (in the second pass of the two-pass step, whose envs are already beyond the first pass's budget: from the first sim step of the pass on; BP_BD_CYCLE=<n>: only
    //  once a path has run n sim steps)
    B.cycle_skip = getenv("BP_BD_CYCLE") ? std::max(0, atoi(getenv("BP_BD_CYCLE"))) : 1;
    // two-pass step: sim steps every env gets in pass 0 (the mean env needs ~750, the 99th percentile ~3 000; the reference's loops stop at 10 001); BP_BD_BUDGET=0: one pass
    h->bd_budget = getenv("BP_BD_BUDGET") ? atoi(getenv("BP_BD_BUDGET")) : 3000;
    const int NW = B.SH * B.SW, words = (NW + 31) / 32, nm = (int)h->bd_maps.size();
    BdPtrs &Q = h->Q;
    int *d_mot; unsigned *d_free, *d_thin; unsigned short *d_edt; float *d_rec; unsigned char *d_small, *d_rchan; d2 *d_rp, *d_rn;
    if ((rc = dalloc(h, &d_mot, T))) return rc;
    if ((rc = dalloc(h, &d_free, (size_t)nm * words))) return rc;
    if ((rc = dalloc(h, &d_thin, (size_t)nm * words))) return rc;
    if ((rc = dalloc(h, &d_edt, (size_t)nm * NW * 2))) return rc;
    if ((rc = dalloc(h, &d_rec, (size_t)nm * NW))) return rc;
    if ((rc = dalloc(h, &d_small, (size_t)nm * NW))) return rc;
    if ((rc = dalloc(h, &d_rp, (size_t)nm * 4))) return rc;
    if ((rc = dalloc(h, &d_rn, (size_t)nm * 4))) return rc;
    if ((rc = dalloc(h, &d_rchan, (size_t)B.local_px * B.local_px))) return rc;
    HIPCHK(h, hipMemcpy(d_mot, h->bd_map_of_trial.data(), sizeof(int) * T, hipMemcpyHostToDevice));
    std::vector<int> first_trial_of_map(nm, -1);
    for (int t = 0; t < T; t++) if (first_trial_of_map[h->bd_map_of_trial[t]] < 0) first_trial_of_map[h->bd_map_of_trial[t]] = t;
    for (int m = 0; m < nm; m++) {
        const BdMaps &M = h->bd_maps[m];
        HIPCHK(h, hipMemcpy(d_free + (size_t)m * words, M.free_bits.data(), sizeof(unsigned) * words, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(d_thin + (size_t)m * words, M.thin_bits.data(), sizeof(unsigned) * words, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(d_edt + (size_t)m * NW * 2, M.edt.data(), sizeof(unsigned short) * NW * 2, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(d_rec + (size_t)m * NW, M.recept.data(), sizeof(float) * NW, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(d_small + (size_t)m * NW, M.small_free.data(), NW, hipMemcpyHostToDevice));
        if (cf.task == 0) {
            const Shape &r = recepts[first_trial_of_map[m]];
            const std::vector<P2> wv = world_verts(r);
            d2 hp[4], hn[4];
            for (int i = 0; i < 4; i++) { hp[i].x = wv[i].x; hp[i].y = wv[i].y; hn[i].x = r.normals[i].x; hn[i].y = r.normals[i].y; }
            HIPCHK(h, hipMemcpy(d_rp + (size_t)m * 4, hp, sizeof(hp), hipMemcpyHostToDevice));
            HIPCHK(h, hipMemcpy(d_rn + (size_t)m * 4, hn, sizeof(hn), hipMemcpyHostToDevice));
        }
    }
    {   // robot_state_channel (box_delivery_env.py:124-131) * 255
        const int lp = B.local_px;
        std::vector<unsigned char> rc_((size_t)lp * lp, 0);
        const int rpw = (int)(2 * cf.robot_radius * cf.ppm);
        const int start = (int)std::floor((double)lp / 2 - (double)rpw / 2);
        for (int i = start; i < start + rpw; i++)
            for (int j = start; j < start + rpw; j++) {
                if (i < 0 || j < 0 || i >= lp || j >= lp) continue;
                const double a = ((double)i + 0.5) - (double)lp / 2, b = ((double)j + 0.5) - (double)lp / 2;
                if (std::sqrt(a * a + b * b) < (double)rpw / 2) rc_[(size_t)i * lp + j] = 255;
            }
        HIPCHK(h, hipMemcpy(d_rchan, rc_.data(), rc_.size(), hipMemcpyHostToDevice));
    }
    Q.map_of_trial = d_mot; Q.free_bits = d_free; Q.thin_bits = d_thin; Q.edt = d_edt; Q.recept = d_rec; Q.small_free = d_small;
    Q.recept_poly = d_rp; Q.recept_n = d_rn; Q.robot_chan = d_rchan;
    const size_t E = (size_t)h->num_envs + (size_t)T;
    if ((rc = dalloc(h, &Q.alive, E * BD_MAXBOX))) return rc;
    if ((rc = dalloc(h, &Q.order, E * BD_MAXBOX))) return rc;
    if ((rc = dalloc(h, &Q.nalive, E))) return rc;
    if ((rc = dalloc(h, &Q.nprev, E))) return rc;
    if ((rc = dalloc(h, &Q.boxdist, E * BD_MAXBOX))) return rc;
    if ((rc = dalloc(h, &Q.boxpos, E * BD_MAXBOX))) return rc;
    if ((rc = dalloc(h, &Q.prev, E * BD_MAXBOX * 4))) return rc;
    if ((rc = dalloc(h, &Q.cum, E * 4))) return rc;
    if ((rc = dalloc(h, &Q.cnt, E * 4))) return rc;
    if ((rc = dalloc(h, &Q.wp, E * BD_MAXWP * 3))) return rc;
    if ((rc = dalloc(h, &Q.nwp, E))) return rc;
    if ((rc = dalloc(h, &Q.stepf, E * 8))) return rc;
    if ((rc = dalloc(h, &Q.unfin, E))) return rc;
    if ((rc = dalloc(h, &Q.rs_i, E * 16))) return rc;
    if ((rc = dalloc(h, &Q.rs_d, E * 132))) return rc;
    if ((rc = dalloc(h, &Q.straggler, (size_t)4))) return rc;
    if ((rc = dalloc(h, &Q.dist, E * NW))) return rc;
    if ((rc = dalloc(h, &Q.rmap, E * NW))) return rc;
    if ((rc = dalloc(h, &Q.cleared, E * BD_MAXBOX))) return rc;
    {
        d2 *d_goals;
        if ((rc = dalloc(h, &d_goals, 128))) return rc;
        if (cf.task == 1) HIPCHK(h, hipMemcpy(d_goals, cf.goal_points, sizeof(d2) * (size_t)cf.num_goal_points, hipMemcpyHostToDevice));
        Q.goals = d_goals;
    }
    h->bd_lds = (size_t)words * 8 + (size_t)3 * BD_QCAP * 2 + 16 + (size_t)BD_PATHCAP * 2 * 2 + BD_PATHCAP + (size_t)BD_PATHCAP * 4 + (size_t)BD_MAXWP * 16 + 64;
    h->bd_obs_lds = (size_t)((NW + 15) & ~15);
    if (h->bd_lds > 160 * 1024 || h->bd_obs_lds > 160 * 1024) return fail(h, BP_EINVAL, "map window too large for LDS");
    HIPCHK(h, hipFuncSetAttribute((const void *)k_bd_settle, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    HIPCHK(h, hipFuncSetAttribute((const void *)k_bd_physics, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    HIPCHK(h, hipFuncSetAttribute((const void *)k_bd_physics_resume, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    HIPCHK(h, hipFuncSetAttribute((const void *)k_bd_physics_damp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    HIPCHK(h, hipFuncSetAttribute((const void *)k_bd_settle_damp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes));
    h->damp = h->P.damping_pow != 0.0;   // space.damping != 0: the generic instantiations (substep<BP_ENV_BOX, DAMP = true>)
    HIPCHK(h, hipFuncSetAttribute((const void *)k_bd_plan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bd_lds));
    HIPCHK(h, hipFuncSetAttribute((const void *)k_bd_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bd_lds));
    HIPCHK(h, hipFuncSetAttribute((const void *)k_ac_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bd_lds));
    HIPCHK(h, hipFuncSetAttribute((const void *)k_bd_observe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bd_obs_lds));
    {
        const size_t words = (size_t)((h->B.SH * h->B.SW + 31) / 32);
        h->bd_rmap_lds = ((words + 3) & ~(size_t)3) * 4 + (size_t)3 * BD_QCAP * 2 + 16;
        HIPCHK(h, hipFuncSetAttribute((const void *)k_bd_robot_map, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bd_rmap_lds));
    }
    // settle every trial once into its template slot, then the episode-start bookkeeping (box distances, robot map)
    if (h->damp) hipLaunchKernelGGL(k_bd_settle_damp, dim3(T), dim3(64), h->lds_bytes, 0, h->P, h->D, (const unsigned char *)nullptr, (double *)nullptr, 1);
    else hipLaunchKernelGGL(k_bd_settle, dim3(T), dim3(64), h->lds_bytes, 0, h->P, h->D, (const unsigned char *)nullptr, (double *)nullptr, 1);
    HIPCHK(h, hipGetLastError());
    if (cf.task == 1)
        hipLaunchKernelGGL(k_ac_finish, dim3(T), dim3(64), h->bd_lds, 0, h->P, h->D, h->B, h->Q, 1, 1, (double *)nullptr, (unsigned char *)nullptr,
                           (unsigned char *)nullptr, (double *)nullptr);
    else
        hipLaunchKernelGGL(k_bd_finish, dim3(T), dim3(64), h->bd_lds, 0, h->P, h->D, h->B, h->Q, 1, 1, (double *)nullptr, (unsigned char *)nullptr,
                           (unsigned char *)nullptr, (double *)nullptr);
    HIPCHK(h, hipGetLastError());
    hipLaunchKernelGGL(k_bd_robot_map, dim3(T), dim3(BDR_THREADS), h->bd_rmap_lds, 0, h->P, h->D, h->B, h->Q, 1);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipDeviceSynchronize());
    h->loaded = true;
    return BP_OK;
}

int bp_bd_get_maps(bp_handle *h, int32_t trial, int32_t *dims, uint8_t *cspace, uint8_t *cspace_thin, uint16_t *edt, float *recept, uint8_t *small_free)
{
    if (!h) return BP_EINVAL;
    if (!h->loaded || h->P.env_kind != BP_ENV_BOX) return fail(h, BP_ESTATE, "not a loaded box-delivery handle");
    if (trial < 0 || trial >= h->num_trials) return BP_EINVAL;
    const bpgeom::BdMaps &M = h->bd_maps[h->bd_map_of_trial[trial]];
    if (dims) { dims[0] = M.H; dims[1] = M.W; dims[2] = M.SH; dims[3] = M.SW; dims[4] = M.si0; dims[5] = M.sj0; }
    const int NW = M.SH * M.SW;
    for (int w = 0; w < NW; w++) {
        if (cspace) cspace[w] = (M.free_bits[w >> 5] >> (w & 31)) & 1u;
        if (cspace_thin) cspace_thin[w] = (M.thin_bits[w >> 5] >> (w & 31)) & 1u;
    }
    if (edt) memcpy(edt, M.edt.data(), sizeof(uint16_t) * NW * 2);
    if (recept) memcpy(recept, M.recept.data(), sizeof(float) * NW);
    if (small_free) memcpy(small_free, M.small_free.data(), NW);
    return BP_OK;
}

int bp_bd_get_state(bp_handle *h, uint8_t *alive, double *waypoints, int32_t *nwp)
{
    if (!h) return BP_EINVAL;
    if (!h->loaded || h->P.env_kind != BP_ENV_BOX) return fail(h, BP_ESTATE, "not a loaded box-delivery handle");
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    const size_t E = h->num_envs;
    if (alive) HIPCHK(h, hipMemcpy(alive, h->Q.alive, E * BD_MAXBOX, hipMemcpyDeviceToHost));
    if (waypoints) HIPCHK(h, hipMemcpy(waypoints, h->Q.wp, sizeof(double) * E * BD_MAXWP * 3, hipMemcpyDeviceToHost));
    if (nwp) HIPCHK(h, hipMemcpy(nwp, h->Q.nwp, sizeof(int) * E, hipMemcpyDeviceToHost));
    return BP_OK;
}

int bp_get_num_bodies(bp_handle *h, int32_t *out_host)
{
    if (!h || !out_host) return BP_EINVAL;
    if (!h->loaded) return fail(h, BP_ESTATE, "not loaded");
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out_host, h->D.e_nb, sizeof(int) * h->num_envs, hipMemcpyDeviceToHost));
    return BP_OK;
}

int bp_check_errors(bp_handle *h, int32_t *out_host)
{
    if (!h) return BP_EINVAL;
    if (!h->loaded) return fail(h, BP_ESTATE, "not loaded");
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    std::vector<int> e(h->num_envs);
    HIPCHK(h, hipMemcpy(e.data(), h->D.e_err, sizeof(int) * h->num_envs, hipMemcpyDeviceToHost));
    int any = 0;
    for (int v : e) any |= v;
    if (out_host) memcpy(out_host, e.data(), sizeof(int) * h->num_envs);
    if (any) return fail(h, BP_ECAPACITY, "in-kernel capacity overflow, bits=" + std::to_string(any));
    return BP_OK;
}

int bp_get_step_cycles(bp_handle *h, uint32_t *out_host)
{
    if (!h || !out_host) return BP_EINVAL;
    if (!h->loaded) return fail(h, BP_ESTATE, "not loaded");
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out_host, h->D.e_cost, sizeof(unsigned) * h->num_envs, hipMemcpyDeviceToHost));
    return BP_OK;
}

int32_t bp_sched_chunk(bp_handle *h) { return h ? h->sched_chunk : 0; }
int32_t bp_sched_resident(bp_handle *h)
{
    if (!h || h->sched_chunk <= 0 || (h->resident_auto && h->device_shared)) return 0;
    return h->P.pair_mode == 2 ? h->pair_resident : h->sched_persist;
}
int32_t bp_device_shared(bp_handle *h) { return (h && h->resident_auto && h->device_shared) ? 1 : 0; }
int32_t bp_bd_budget(bp_handle *h) { return h ? h->bd_budget : 0; }
int bp_launch_policy_query(int32_t num_envs, int32_t num_compute_units, int32_t can_pair, int32_t is_maze, int32_t *out8_host)
{
    if (!out8_host || num_envs <= 0 || num_compute_units <= 0) return BP_EINVAL;
    const BpLaunchPolicy p = bp_launch_policy(num_envs, num_compute_units, can_pair != 0, is_maze != 0);
    out8_host[0] = p.wave_slots; out8_host[1] = p.pair_mode; out8_host[2] = p.tight; out8_host[3] = p.pair_solo; out8_host[4] = p.chunk;
    out8_host[5] = p.pp_max_act; out8_host[6] = p.pp_max_work; out8_host[7] = p.pp_rate;
    return BP_OK;
}

int bp_get_clock_stamps(bp_handle *h, uint64_t *out16_host)
{
    uint64_t *out2_host = out16_host;
    if (!h || !out2_host) return BP_EINVAL;
    if (!h->loaded) return fail(h, BP_ESTATE, "not loaded");
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out2_host, h->D.clk, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return BP_OK;
}

int32_t bp_pair_mode(bp_handle *h) { return h ? h->pair_mode : 0; }

int bp_bd_get_stragglers(bp_handle *h, uint32_t *out2_host)
{
    if (!h || !out2_host) return BP_EINVAL;
    if (!h->loaded || h->P.env_kind != BP_ENV_BOX || h->Q.straggler == nullptr) return fail(h, BP_ESTATE, "not a loaded box-delivery / area-clearing handle");
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out2_host, h->Q.straggler, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BP_OK;
}
int bp_bd_get_cycle_skips(bp_handle *h, uint32_t *out2_host)
{
    if (!h || !out2_host) return BP_EINVAL;
    if (!h->loaded || h->P.env_kind != BP_ENV_BOX || h->Q.straggler == nullptr) return fail(h, BP_ESTATE, "not a loaded box-delivery / area-clearing handle");
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out2_host, h->Q.straggler + 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return BP_OK;
}

int bp_get_pair_stats(bp_handle *h, int32_t *out16_host)
{
    if (!h || !out16_host) return BP_EINVAL;
    if (!h->loaded) return fail(h, BP_ESTATE, "not loaded");
    BP_DEVICE(h);
    const DevParams &P = h->P;
    const int32_t par[8] = {h->pair_mode, P.pair_solo, P.pp_max_keys, P.pp_max_slots, P.pp_max_mv, P.pp_max_act, P.pp_max_work, P.pp_rate};
    memcpy(out16_host, par, sizeof(par));
    memset(out16_host + 8, 0, 8 * sizeof(int32_t));
    if (h->D.sq_pairstat != nullptr) {
        HIPCHK(h, hipDeviceSynchronize());
        HIPCHK(h, hipMemcpy(out16_host + 8, h->D.sq_pairstat, 8 * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return BP_OK;
}

int bp_sched_warnings(bp_handle *h, int32_t *out2_host)
{
    if (!h || !out2_host) return BP_EINVAL;
    out2_host[0] = out2_host[1] = 0;
    if (h->sched_chunk <= 0 || !h->D.sq_warn) return BP_OK;
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out2_host, h->D.sq_warn, 2 * sizeof(int), hipMemcpyDeviceToHost));
    return BP_OK;
}

int bp_set_step_cost_hint(bp_handle *h, const uint32_t *host_costs)
{
    if (!h || !host_costs) return BP_EINVAL;
    if (!h->loaded) return fail(h, BP_ESTATE, "not loaded");
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(h->D.e_cost, host_costs, sizeof(unsigned) * h->num_envs, hipMemcpyHostToDevice));
    return BP_OK;
}

int bp_enable_timing(bp_handle *h, int32_t on)
{
    if (!h) return BP_EINVAL;
    h->timing = on != 0;
    h->ev_used = 0;
    if (on && h->loaded && h->P.env_kind != BP_ENV_BOX) {   // per-launch cost statistics ride along (bp_get_cost_stats)
        BP_DEVICE(h);
        if (!h->cost_ring) { int rc = dalloc(h, &h->cost_ring, (size_t)BP_COST_RING * 2); if (rc) return rc; }
        h->cost_n = 0;
    }
    return BP_OK;
}

int bp_get_cost_stats(bp_handle *h, uint64_t *out_host, int32_t max_launches, int32_t *launches)
{
    if (!h || !launches || (max_launches > 0 && !out_host)) return BP_EINVAL;
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    const int n = std::min(h->cost_n, std::max(0, (int)max_launches));
    if (n > 0) HIPCHK(h, hipMemcpy(out_host, h->cost_ring, (size_t)n * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    *launches = n;
    return BP_OK;
}

int bp_kernel_time_ms(bp_handle *h, double *physics_ms, double *raster_ms, int32_t *launches)
{
    if (!h) return BP_EINVAL;
    BP_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    double p = 0, r = 0;
    const int n = (int)(h->ev_used / 3);
    for (int k = 0; k < n; k++) {
        float a = 0, b = 0;
        HIPCHK(h, hipEventElapsedTime(&a, h->ev[3 * k], h->ev[3 * k + 1]));
        HIPCHK(h, hipEventElapsedTime(&b, h->ev[3 * k + 1], h->ev[3 * k + 2]));
        p += a; r += b;
    }
    if (physics_ms) *physics_ms = n ? p / n : 0.0;
    if (raster_ms) *raster_ms = n ? r / n : 0.0;
    if (launches) *launches = n;
    h->ev_used = 0;
    return BP_OK;
}

// reset() normally copies the per-trial settled template (a pure function of the trial); on != 0 makes it re-run the
// 1000 settle sub-steps in place like the reference does (same result; used by the tests to prove that).
int bp_set_resettle(bp_handle *h, int32_t on)
{
    if (!h) return BP_EINVAL;
    h->resettle = (on != 0) || (h->P.random_start != 0); // per-episode start poses always settle in place
    return BP_OK;
}

// debug hook of the diagnostic twin (-DBP_DEBUG_PATHS): trace (x, y, angle) of every body of one env after each sub-step of the next
// launches into a device buffer [substeps][nb_cap][3]; pass NULL to disable.
int bp_debug_trace(bp_handle *h, double *dev_buf, int32_t env)
{
    if (!h) return BP_EINVAL;
#ifndef BP_DEBUG_PATHS
    // the product kernels do not test the trace pointer every sub-step; the diagnostic twin does (benchpush_amd/build.py: build_debug_paths)
    if (dev_buf) return fail(h, BP_ESTATE, "bp_debug_trace needs the -DBP_DEBUG_PATHS build (BP_PROF=1 BP_PROF_LIB=.../libbenchpush_hip_dbgpaths.so)");
#endif
    h->D.dbg = dev_buf;
    h->D.dbg_env = env;
    return BP_OK;
}

// diagnostic builds (-DBP_PROF): per-env phase cycle counters, device buffer [E][24] u64 or NULL
int bp_debug_prof(bp_handle *h, unsigned long long *dev_buf)
{
    if (!h) return BP_EINVAL;
    h->D.prof = dev_buf;
    return BP_OK;
}

} // extern "C"
