// cs_api.hip -- host side of the C ABI declared in include/clearsky_hip.h (context, uploads, launches).
// Reference interfaces replaced are cited in the header; orchestration follows fluxes.jl:238-279 / :357-383.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/clearsky_hip.h"
#include "../../include/clearsky_hip_dev.h"
#include "cs_kernels.h"

using namespace csdev;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

thread_local int g_nlaunch = 0;   // kernel launches since the last reset (cs_column_run reports its count)
thread_local int g_near_launches = 0, g_line_kernel = 0;   // of the step being enqueued: near-line launches (all groups), and what summed the per-point
                                                           // far lines of its last group: 0 = k_voigt_far, 1 = k_linesum<shape>, 2 = k_phco2
#define CS_LAUNCH(...) do { g_nlaunch++; hipLaunchKernelGGL(__VA_ARGS__); } while (0)

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(CS_EHIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// owns one device allocation; move-only, so that `x = T()` frees what x held and containers of structs with DevBuf members can
// grow without double frees
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t reserve(size_t n)
    {
        if (n <= bytes && p) return hipSuccess;
        release();
        if (n == 0) n = 8;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

template <class T> int upload(DevBuf &b, const T *h, size_t n, hipStream_t s)
{
    HIPCHK(b.reserve(n * sizeof(T)));
    if (n) HIPCHK(hipMemcpyAsync(b.p, h, n * sizeof(T), hipMemcpyHostToDevice, s));
    return CS_OK;
}

struct GasTable {
    bool present = false;
    uint64_t generation = 0;   // bumped by every cs_gas_upload into the slot (a resident column's windows belong to one generation)
    int64_t L = 0;
    int niso = 0;
    double mu_min = 0.0, mu_max = 0.0, ga_max = 0.0, gs_max = 0.0, na_min = 0.0, na_max = 0.0;
    // host mirror of the table (58 B per line): what a column merges when several of its gases share one launch set
    std::vector<double> h_nu, h_S, h_ga, h_gs, h_Epp, h_na, h_mu, h_cheb;
    std::vector<int16_t> h_iso;
    std::vector<int32_t> h_ncheb;
    std::vector<uint8_t> h_gid;                          // merged tables only: member index of each line
    std::vector<std::pair<int, uint64_t>> members;       // merged tables only: (slot, generation) of each member, in member order
    DevBuf nu, S, ga, gs, Epp, na, mu, iso, ncheb, cheb, sref, gid;
    GasDev dev() const
    {
        GasDev g;
        g.L = L;
        g.nu = nu.as<double>(); g.S = S.as<double>(); g.ga = ga.as<double>(); g.gs = gs.as<double>();
        g.Epp = Epp.as<double>(); g.na = na.as<double>(); g.mu = mu.as<double>();
        g.iso = iso.as<int16_t>(); g.ncheb = ncheb.as<int32_t>(); g.cheb = cheb.as<double>();
        g.sref = sref.as<double>();
        g.gid = h_gid.empty() ? nullptr : gid.as<uint8_t>();
        return g;
    }
};

struct TableDev {
    bool present = false;
    uint64_t generation = 0;   // bumped whenever the slot is (re)filled (cs_bake, cs_table_upload) or cleared: a resident column's weights
                               // W [nT * nP][K] belong to one generation's (T, P) grid
    int64_t nnu = 0;
    int nT = 0, nP = 0;
    std::vector<double> T, lnP, nu;
    DevBuf Z;  // [nT*nP][nnu] ln sigma
};

struct ColTab {
    int slot = 0;
    uint64_t generation = 0;   // of the table slot the weights were formed for
    DevBuf W, conc;  // [M][K], [K]
};

// AcceleratedAbsorber (absorbers.jl:114-203): ln sigma on pressure knots, per wavenumber
struct AccelDev {
    bool present = false;
    uint64_t generation = 0;   // bumped whenever the KNOTS of the slot change (cs_accel_upload; cs_accel_store with other knots) or it is
                               // cleared: a resident column's knot cells belong to one generation (new VALUES on the same knots keep it)
    int64_t nnu = 0;
    int nk = 0;
    std::vector<double> lnP, nu;   // knots (ascending), wavenumbers
    DevBuf L;                      // [nk][nnu]
};
struct ColAccel {
    int slot = -1;                 // -1: none
    uint64_t generation = 0;       // of the accelerated-absorber slot the cells were formed for
    DevBuf cell, x, xa, xb;        // per node: knot interval and the three abscissae of the LinearInterpolator formula
};

struct CiaBandHost {
    std::vector<double> nu, T;
    DevBuf dnu, dlnk;
};
struct CiaDev {
    bool present = false;
    uint64_t generation = 0;   // bumped whenever the slot's bands change (a resident column's per-grid tables belong to one generation)
    std::vector<CiaBandHost> bands;
    int filled = 0;
};
struct ColCia {
    int slot = 0, flags = 0;
    DevBuf bands, st, rho1, rho2, rhoa;
    DevBuf fac;           // k_cia_tab: [K] Lo^2 rho1 rho2 / rhoa
    DevBuf tab, toff;     // k_cia_tab: ln k of every band at the temperature of every node state, [toff[b] + k * nb + c]
    DevBuf tband, cell, fx;   // k_flux: per 64-point tile the bands that reach it [ntile][CS_CIA_ACT] (-1: none), and per (slot, wavenumber)
                              // the sample cell of that band (-1: outside it) and the position in the cell -- the grid's, not the state's
    uint64_t generation = 0;  // of the CIA slot these per-grid tables were built from
    int nband = 0;
    int max_overlap = 0;  // bands of this object reaching one 64-point tile of the column's grid, at most (k_flux holds CS_CIA_ACT)
};

// interpolated far wings: interval levels of a nu grid (descending interval size; intervals of all levels form one list)
struct ChebGrid {
    int nlev = 0, nItot = 0;
    int itv[CS_MAX_LEVEL] = {}, nI[CS_MAX_LEVEL] = {}, ioff[CS_MAX_LEVEL] = {};
    double span[CS_MAX_LEVEL] = {};   // widest interval of the level (cheb_build; 0: not known)
    DevBuf nodes;               // [nItot][64]
    DevBuf Cm[CS_MAX_LEVEL];    // [nI][64][itv]
    DevBuf Rc[CS_MAX_LEVEL];    // level l >= 1: [nI][64][64], the parent's node values carried to the nodes of an interval (k_cheb_cascade)
    DevBuf tnodes, tC;          // per 64-point tile: 16 Chebyshev nodes [tiles][16] and the matrix that carries values at them to the tile's points
                                // [tiles][16][64] (k_voigt_edge_mx: the window-end lines inside the cut-off of every point of the tile)
    bool tile_nodes_ok = false; // 16 nodes carry such lines to rounding on this grid (far_node_count of their distance; cheb_build)
};
// per gas on that grid: windows per level, zones [K][nItot], node sums F [nItot][64][Kpad]
struct GasInterp { int nlev = 0, l0 = 0; int nfar[CS_MAX_LEVEL] = {}; DevBuf iwin[CS_MAX_LEVEL], iz, F, sep, edge; };   // nfar: nodes for a level's far pieces (far_node_count)   // levels l0 .. nlev-1 of the grid are in use; sep: SepZone [K/16][nItot]
                                                                                          // (matrix-core node sums), edge: EdgeZone [K/16][tiles] (matrix-core pieces of the per-point sum)

// a gas of the column as the caller named it (conc is laid out [ngas, K] over these)
struct UserGas {
    int slot = 0, shape = 0;
    double cut = 25.0;
    uint64_t generation = 0;
    int64_t pairs_per_state = -1, lines_in_range = 0;
};
// a launch group: one gas, or all Voigt (Lorentz) gases of the column with the same cut-off merged into ONE sorted line table --
// sigma_total = sum_g C_g sigma_g (absorbers.jl:84-95) and a per-(state, line) record carries everything gas-specific (the
// member's concentration and partial pressure enter in k_gas_setup), so the kernels see one table: one launch set per column
// instead of one per gas, and windows as dense as the column's lines together
struct ColGas {
    std::vector<int> mem;     // indices into Column::ugas
    const GasTable *tab = nullptr;   // ctx->gas[slot], or a merged table ...
    std::shared_ptr<const GasTable> hold;   // ... which the column shares with the context's cache (an eviction there cannot free it)
    int shape = 0;
    double cut = 25.0;
    DevBuf conc, Pp, J0, J1;  // [nmem][K], [nmem][K], [ntile], [ntile]
    DevBuf lrt, qref;         // state_tables(): [K], [K][niso]
    DevBuf win, zones, gmax;  // [ntile64] WaveWin, [K][ntile64] Zone, [K] max Lorentz width (Voigt fast path)
    GasInterp itp;            // interpolated far wings (nlev = 0: off)
    int64_t jlo = 0, jhi = 0;
    int xtiles = 0;           // longest XCD stretch of the far kernel's tile order, in tiles (wave_windows)
};

// k_rt launch geometry (rt_geometry)
struct RtGeom { bool ud, streams; int tiles, nblk, threads; size_t shmem; };   // streams: k_rt_streams (one wave per stream and sweep)

struct Column {
    double *F_dst = nullptr;   // cs_column_set_flux_dst: caller-owned device memory the band fluxes [2 np] are written to (NULL: the column's own F)
    double *flux_out() { return F_dst ? F_dst : F.as<double>(); }
    bool ready = false;
    int64_t nnu = 0;
    int np = 0, nl = 0, nlob = 0, K = 0, nstream = 0, ngas = 0, ntile = 0;
    RtGeom rtg = {};
    bool want_tau = false, want_M = false, has_extra = false, has_S = false, has_alb = false;
    bool default_wts = false;  // trapezoid weights of the column's own grid (not a shard of a larger one)
    int64_t g_nnu = 0, g_start = 0;      // set by cs_fluxes_discretized_multi: this column is points [g_start, g_start + nnu) of a grid of
    double g_left = 0.0, g_right = 0.0;  // g_nnu points, with these neighbours left and right (what its trapezoid weights depend on)
    int interp = 0;            // the context's interpolation settings at setup time (packed)
    double g = 0, sigma_gray = 0, theta_s = 0;
    RtParams rt;
    std::vector<double> h_P, h_Pk, h_xs, h_nu;
    std::vector<UserGas> ugas;   // the caller's gases (ngas of them)
    std::vector<ColGas> gas;     // launch groups
    int merge = 1;               // the context's cs_set_merge at setup time
    uint64_t grid_id = 0;        // names this setup's nu grid (what the PHCO2 path keeps per grid: PhScratch)
    int launches = 0;            // kernel launches of the last cs_column_run
    bool near_live = false;      // the last cs_column_run left its near-line pairs in sigma2 (k_rt read both planes): cs_column_sigma_fetch folds them in
    hipStream_t last_stream = nullptr;   // the stream that run was enqueued on (cs_column_fetch waits for it, not for the whole device)
    // cs_set_tuning key 4: the step as ONE hipGraph launch (captured on the second run after a change, replayed from the third on):
    // kernel arguments are device addresses that stay put between cs_column_update_state calls, so only what changes launch
    // geometry or pointers (setup, tables, CIA pairs, accelerated absorber, spectra switched on or off) drops the graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int runs_since_change = 0;
    bool graph_near_live = false, graph_sigma_partial = false;   // what the captured run left in near_live / sigma_partial / launches: a replay leaves the same
    int graph_launches = 0;
    std::vector<ColTab> tab;
    std::vector<ColCia> cia;
    ColAccel accel;
    std::vector<double> h_Tk;
    DevBuf nu, wts, P, Pk, Tk, muk, Tlev, extra, S_toa, albedo;
    DevBuf hot, cold, sigma, sigma2, tau, Mup, Mdn, partial, F, stage, ranges;   // sigma2: the near-line plane (k_voigt_near on a side stream)
    DevBuf fluxdbg;            // k_flux_scan: phase time stamps of block 0 (cs_set_tuning key 15 | 128; cs_column_work out[27..])
    DevBuf ticket;             // k_flux: blocks finished (the last one adds the block partials up)
    int flux_form_last = 0;      // which flux kernel the last run used (flux_form)
    int near_launches_last = 0, line_kernel_last = 0;   // cs_column_info out[6], out[7]
    bool sigma_partial = false;  // the last run finished the cross-sections on chip (k_flux): cs_column_sigma_fetch evaluates them again, in HBM
    ChebGrid cheb;             // interpolation levels of the nu grid (nlev = 0: off)
    DevBuf chebF;              // node sums F [nItot][64][Kpad], summed over the column's gases (k_cheb_nodes accumulates)
};

}  // namespace

static std::atomic<int> g_live_ctx{0};   // contexts alive in this process (the last cs_destroy stops the worker pool)
static void pool_stop();

static void drop_graph(Column &c)
{
    if (c.graph_exec) (void)hipGraphExecDestroy(c.graph_exec);
    if (c.graph) (void)hipGraphDestroy(c.graph);
    c.graph_exec = nullptr;
    c.graph = nullptr;
    c.runs_since_change = 0;
}

// workspace of the PHCO2 fast path (k_phco2): per-(state, line) chi factors and per-tile region windows
constexpr int CS_NTUNE = 24;
// nu_lo .. grid_id: the grid of the call, set by the caller (ph_set_grid); cheb, piw, F: the interpolation levels of that grid for the
// PHCO2 cut-off (k_phco2_nodes), rebuilt when the key (grid_id, nnu, cut) changes
struct PhScratch {
    DevBuf fac, win, piw, F, win3, zones3;   // win3, zones3: windows and zones of the Voigt pass over the pairs within 3 cm^-1
    struct Dens { const void *tab; uint64_t gen, grid; bool ok; };
    std::vector<Dens> dens;                  // check_near_density() per (table, grid), a few kept
    double nu_lo = 0.0, nu_hi = 0.0, nu_c = 0.0, max_span = 0.0, margin = kChebMargin;
    int itp_on = 0, itp_min = 128, itp_max = 2048, own_core = 0;   // own_core: cs_set_tuning key 9
    uint64_t grid_id = 0;
    double lev_span[8] = {};   // widest interval of 8192 >> i points on this grid (ph_set_grid)
    int force64 = 0;           // cs_set_tuning key 10: 64 nodes for every interval (bit 0), tiles as 64-point intervals too (bit 1)
    // interval sizes in use (descending) and the virtual levels on them (PhVLevels); nodes [sum nI x nc], Cm[v] [nI][nc][itv]
    struct Grid {
        PhLevels lv; PhVLevels vl; PhFine fine;
        int nnodes = 0, nslots = 0;
        DevBuf nodes, Cm[CS_MAX_ALEVEL];
    } grid;
    uint64_t built_id = 0; int64_t built_nnu = 0; double built_cut = 0.0, built_margin = 0.0; int built_min = 0, built_max = 0, built_force = 0;
};

// what cs_fluxes_discretized_multi keeps between calls (in its first context)
struct MultiPlan { int nctx = 0; std::vector<double> nu, wt; std::vector<uint64_t> gens; std::vector<int64_t> ranges; };

struct cs_ctx {
    PhScratch ph;
    MultiPlan mplan;
    int device = 0;
    bool counted = false;   // cs_create finished: the context counts in g_live_ctx
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;            // side stream: node sums beside the per-point kernels (cs_set_tuning key 2)
    hipStream_t stream3 = nullptr;            // side stream: near-line kernels beside the matrix-core per-point kernel (key 7)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork3 = nullptr, ev_join3 = nullptr, ev_far3 = nullptr;
    GasTable gas[CS_MAX_GAS];
    TableDev tab[CS_MAX_TABLE];
    CiaDev cia[CS_MAX_CIA];
    AccelDev accel[CS_MAX_ACCEL];
    Column col;
    int mixed = 0;
    int interp = 1;   // far wings by Chebyshev interpolation over 128..2048-point intervals (k_cheb_nodes / k_cheb_apply)
    int itp_first = -1, itp_min = 128, itp_max = 2048;   // cs_set_interp_plan: first level per gas (-1 = by line density), size range
    int matrix_nodes = 1;   // cs_set_matrix_cores: separable far-wing node sums on v_mfma_f64 (k_cheb_nodes_mx)
    int matrix_core = 1;    // ... and the window core on sub-tiles (k_voigt_sub + the second mask of k_voigt_edge_mx)
    int merge = 1;          // cs_set_merge: gases of a column with the same shape and cut-off share one merged line table
    // cs_set_tuning (include/clearsky_hip.h has the full descriptions): [0] interpolated wings applied inside k_voigt_edge_mx, [1]
    // matrix-core kernels on short grids, [2] node sums on a side stream, [3] interpolation margin (per cent), [4] hipGraph replay, [5]
    // k_rt_streams on short grids, [6] split levels of k_cheb_nodes_mx, [7] near-line kernels on a second side stream, [8] states of a
    // group that must be able to use a line for it to join the group's matrix-core node piece, [9] PHCO2 core in k_phco2 itself, [10]
    // PHCO2 node counts / 64-point intervals, [11] far pieces of the node sums on all 64 nodes, [12] level cascade, [13] k_cheb_nodes
    // with four waves per (interval, state), [14] cut-off edges of k_voigt_edge_mx without the sub-tile phases, [15] the flux kernel
    // finishes the cross-sections on chip (k_flux): 0 = where it pays, 1 = never, 2 = always
    bool gfx950 = false;    // cs_create: the device reports gcnArchName gfx950 (the in-kernel band sum of k_flux_* is used only then)
    int tune[CS_NTUNE] = {0, 1, 2, 0, 0, 1, 0, 1, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<std::shared_ptr<GasTable>> merged;   // merged tables (keyed by their members' (slot, generation)), least recently used first
    double far_s = 1e6;
    DevBuf reinterp;        // [64][32] then [64][16]: values at the 64 nodes of an interval from those at its 32 / 16 nodes (build_reinterp)
    DevBuf hot32;
    DevBuf tmpA, tmpB, tmpC;
};
void ph_set_grid(const cs_ctx *ctx, PhScratch &ph, const double *nu, int64_t nnu, uint64_t grid_id)
{
    ph.nu_lo = nu[0]; ph.nu_hi = nu[nnu - 1];
    ph.max_span = 0.0;
    for (int64_t i0 = 0; i0 < nnu; i0 += 64) ph.max_span = std::max(ph.max_span, nu[std::min(i0 + 63, nnu - 1)] - nu[i0]);
    ph.grid_id = grid_id;
    ph.itp_on = ctx->interp; ph.itp_min = ctx->itp_min; ph.itp_max = ctx->itp_max;
    ph.margin = ctx->tune[3] > 0 ? 0.01 * ctx->tune[3] : kChebMargin;
    ph.own_core = ctx->tune[9];
    ph.force64 = ctx->tune[10];
    for (int i = 0; i < 8; i++) {
        const int64_t sz = 8192 >> i;
        double w = 0.0;
        for (int64_t i0 = 0; i0 < nnu; i0 += sz) w = std::max(w, nu[std::min(i0 + sz - 1, nnu - 1)] - nu[i0]);
        ph.lev_span[i] = w;
    }
}

namespace {

// Gauss-Legendre / Gauss-Lobatto rules on [-1,1] (FastGaussQuadrature.gausslegendre / gausslobatto are the
// mathematically unique rules), ascending nodes.  Newton on P_n with the three-term recurrence.
void legendre(int n, double z, double &pn, double &pnm1)
{
    double p0 = 1.0, p1 = z;
    if (n == 0) { pn = 1.0; pnm1 = 0.0; return; }
    for (int j = 1; j < n; j++) {
        double p2 = ((2.0 * j + 1.0) * z * p1 - j * p0) / (j + 1.0);
        p0 = p1;
        p1 = p2;
    }
    pn = p1;
    pnm1 = p0;
}

void gauss_legendre(int n, double *x, double *w)
{
    for (int i = 0; i < n; i++) {
        double z = std::cos(M_PI * (i + 0.75) / (n + 0.5));
        double pn, pm, dp = 1.0;
        for (int it = 0; it < 64; it++) {
            legendre(n, z, pn, pm);
            dp = n * (z * pn - pm) / (z * z - 1.0);
            double dz = pn / dp;
            z -= dz;
            if (std::fabs(dz) < 1e-16) break;
        }
        legendre(n, z, pn, pm);
        dp = n * (z * pn - pm) / (z * z - 1.0);
        x[n - 1 - i] = z;
        w[n - 1 - i] = 2.0 / ((1.0 - z * z) * dp * dp);
    }
}

void gauss_lobatto(int n, double *x, double *w)
{
    const int N = n - 1;
    x[0] = -1.0;
    x[N] = 1.0;
    w[0] = w[N] = 2.0 / (N * (N + 1.0));
    for (int i = 1; i < N; i++) {
        double z = -std::cos(M_PI * i / N), pn, pm;
        for (int it = 0; it < 64; it++) {
            legendre(N, z, pn, pm);
            double d1 = N * (pm - z * pn) / (1.0 - z * z);
            double d2 = (2.0 * z * d1 - N * (N + 1.0) * pn) / (1.0 - z * z);
            double dz = d1 / d2;
            z -= dz;
            if (std::fabs(dz) < 1e-16) break;
        }
        legendre(N, z, pn, pm);
        x[i] = z;
        w[i] = 2.0 / (N * (N + 1.0) * pn * pn);
    }
}

// Lagrange basis of the interpolating polynomial on Chebyshev extrema x[0..n) (ascending), barycentric form:
// l_i(v) = (w_i/(v-x_i)) / sum_j (w_j/(v-x_j)),  w_i = (-1)^i * (1/2 at the two ends).  This is the unique polynomial
// BichebyshevInterpolator evaluates (gases.jl:80,85); only rounding can differ from the reference's algorithm.
void cheb_basis(const std::vector<double> &x, double v, std::vector<double> &l)
{
    const int n = (int)x.size();
    l.assign(n, 0.0);
    for (int i = 0; i < n; i++)
        if (v == x[i]) { l[i] = 1.0; return; }
    double den = 0.0;
    for (int i = 0; i < n; i++) {
        double w = ((i & 1) ? -1.0 : 1.0) * ((i == 0 || i == n - 1) ? 0.5 : 1.0);
        l[i] = w / (v - x[i]);
        den += l[i];
    }
    for (int i = 0; i < n; i++) l[i] /= den;
}

template <int SHAPE>
void launch_linesum(dim3 grid, hipStream_t s, const double *nu, int64_t nnu, int64_t L, const LineHot *hot,
                    const LineCold *cold, const int32_t *J0, const int32_t *J1, double cut, const double *Tk,
                    double base, const double *extra, double *sigma, int accumulate)
{
    CS_LAUNCH(k_linesum<SHAPE>, grid, dim3(256), 0, s, nu, nnu, L, hot, cold, J0, J1, cut, Tk, base, extra,
                       sigma, accumulate);
}

void launch_linesum_shape(int shape, dim3 grid, hipStream_t s, const double *nu, int64_t nnu, int64_t L,
                          const LineHot *hot, const LineCold *cold, const int32_t *J0, const int32_t *J1, double cut,
                          const double *Tk, double base, const double *extra, double *sigma, int accumulate)
{
    switch (shape) {
    case SH_LORENTZ: launch_linesum<SH_LORENTZ>(grid, s, nu, nnu, L, hot, cold, J0, J1, cut, Tk, base, extra, sigma, accumulate); break;
    case SH_DOPPLER: launch_linesum<SH_DOPPLER>(grid, s, nu, nnu, L, hot, cold, J0, J1, cut, Tk, base, extra, sigma, accumulate); break;
    case SH_PHCO2: launch_linesum<SH_PHCO2>(grid, s, nu, nnu, L, hot, cold, J0, J1, cut, Tk, base, extra, sigma, accumulate); break;
    default: launch_linesum<SH_VOIGT>(grid, s, nu, nnu, L, hot, cold, J0, J1, cut, Tk, base, extra, sigma, accumulate); break;
    }
}

// k_rt launch geometry: up/down split (two waves per 64-point tile) while the grid has fewer than ~4 waves per SIMD,
// tiles per block so that the grid still covers the chip
RtGeom rt_geometry(int64_t nnu, int np, int ncol, int ns = 0, bool allow_streams = false)
{
    RtGeom g;
    const int64_t nwave = (nnu + 63) / 64 * ncol;
    g.ud = nwave < 4096;
    g.streams = false;
    g.tiles = g.ud ? (nnu >= 65536 ? 2 : 1) : (nnu >= 65536 ? 4 : 1);
    g.nblk = (int)((nnu + (int64_t)g.tiles * 64 - 1) / ((int64_t)g.tiles * 64));
    g.threads = g.tiles * 64 * (g.ud ? 2 : 1);
    g.shmem = ((size_t)2 * np * g.tiles + (g.ud ? (size_t)g.tiles * 64 : 0)) * sizeof(double);
    // short grids: the sweeps are latency chains -- one wave per (sweep, stream) of a tile instead of one per sweep (k_rt_streams): 1/8
    // of C3 (196 tiles) 0.061 -> 0.046 ms, C2 (157) 0.042 -> 0.033; from ~400 tiles on the two-wave form is as fast or faster (1/4 of
    // C3 0.082 vs 0.085, 1/2 0.087 vs 0.126: ten waves per tile stop fitting beside each other)
    const size_t sh2 = ((size_t)(2 * np - 1) * 64 + (size_t)4 * ns * 64 + (size_t)2 * np + 64) * sizeof(double);   // Planck + optical depths + exchange
    if (allow_streams && g.ud && g.tiles == 1 && nwave <= 400 && ns >= 2 && ns <= 8 && sh2 <= 160 * 1024 - 4096) {
        g.streams = true;
        g.threads = 2 * ns * 64;
        g.shmem = sh2;
    }
    return g;
}

template <int NS>
void launch_rt_ns(const RtGeom &g, int B, hipStream_t s, const RtParams &p, const double *nu, const double *wts,
                  int64_t nnu, const double *sigma, const double *muk, const double *P, const double *Tlev,
                  const double *S, const double *alb, double *tau, double *Mup, double *Mdn, double *partial, size_t sig_bstride,
                  const double *sigma2)
{
    if (g.streams) {
        if constexpr (NS >= 2 && NS <= 8) {
            if (g.shmem > 65536) (void)hipFuncSetAttribute((const void *)k_rt_streams<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.shmem);
            CS_LAUNCH((k_rt_streams<NS>), dim3(g.nblk, B), dim3(g.threads), g.shmem, s, p, nu, wts, nnu, sigma, muk, P, Tlev, S, alb, tau, Mup,
                      Mdn, partial, sig_bstride, sigma2);
        }
    } else if (g.ud)
        CS_LAUNCH((k_rt<NS, true>), dim3(g.nblk, B), dim3(g.threads), g.shmem, s, p, nu, wts, nnu, sigma, muk, P, Tlev, S, alb,
                           tau, Mup, Mdn, partial, sig_bstride, sigma2);
    else
        CS_LAUNCH((k_rt<NS, false>), dim3(g.nblk, B), dim3(g.threads), g.shmem, s, p, nu, wts, nnu, sigma, muk, P, Tlev, S, alb,
                           tau, Mup, Mdn, partial, sig_bstride, sigma2);
}

void launch_rt(int ns, const RtGeom &g, int B, hipStream_t s, const RtParams &p, const double *nu, const double *wts,
               int64_t nnu, const double *sigma, const double *muk, const double *P, const double *Tlev,
               const double *S, const double *alb, double *tau, double *Mup, double *Mdn, double *partial, size_t sig_bstride = 0,
               const double *sigma2 = nullptr)
{
#define CS_RT_CASE(N) case N: launch_rt_ns<N>(g, B, s, p, nu, wts, nnu, sigma, muk, P, Tlev, S, alb, tau, Mup, Mdn, partial, sig_bstride, sigma2); break;
    switch (ns) {
        CS_RT_CASE(1) CS_RT_CASE(2) CS_RT_CASE(3) CS_RT_CASE(4) CS_RT_CASE(5) CS_RT_CASE(6) CS_RT_CASE(7) CS_RT_CASE(8)
        CS_RT_CASE(9) CS_RT_CASE(10) CS_RT_CASE(11) CS_RT_CASE(12) CS_RT_CASE(13) CS_RT_CASE(14) CS_RT_CASE(15) CS_RT_CASE(16)
    }
#undef CS_RT_CASE
}

// which form of the flux kernel a step of the resident column uses: 0 = k_rt / k_rt_streams reading finished cross-sections from HBM,
// 3 = k_flux_scan (short grids), 2 = k_flux_chunk (long grids)
int flux_form(const cs_ctx *ctx, const Column &c, size_t *shmem, int *nblk, int *threads)
{
    if ((ctx->tune[15] & 3) == 1 || !c.tab.empty() || c.gas.empty()) return 0;   // (baked tables are added between wings and CIA pairs: own pass)
    for (auto &cc : c.cia)
        if (cc.max_overlap > CS_CIA_ACT) return 0;
    const int np = c.np, ns = c.nstream, K = c.K;
    const size_t lim = 160 * 1024 - 4096;
    const int nt64_ = (int)((c.nnu + 63) / 64);
    // the scan form: grids of up to 1024 tiles -- with a chunk's transmissivities in registers it beats the separate kernels on a half
    // (846 tiles: 1.10 -> 1.08 ms) and a quarter (407 tiles: 0.677 -> 0.610) of the bench column, not on the whole (1563 tiles: 1.94 ->
    // 1.99) -- and, key 15 | 1024, every grid in k_rt's two-waves-per-tile regime (A/B)
    // (key 15 = 2, `always`, keeps the chunked form on mid-size grids: tests)
    const bool scan_ok = c.rtg.streams || (c.rtg.ud && ns >= 2 && ns <= 8 && nt64_ >= 1 && ((nt64_ <= 1024 && (ctx->tune[15] & 3) != 2) || (ctx->tune[15] & 1024)));
    if (scan_ok) {
        // the sweeps as a scan over layer chunks: five layers per wave where 12 waves reach (their transmissivities stay in registers,
        // k_flux_scan<NS, 5>), whole waves per SIMD (4, 8 or 12: ten waves of six layers load two SIMDs with three waves and two with two)
        const int nw = std::min(12, std::max(4, 4 * ((c.nl + 19) / 20)));
        const size_t sh = ((size_t)K * 64 + (size_t)(2 * np - 1) * 64 + (size_t)2 * (ns + 1) * 64 + (size_t)2 * np) * sizeof(double);
        if (sh > lim) return 0;
        *shmem = sh; *nblk = nt64_; *threads = nw * 64;
        return 3;
    }
    // long grids: where k_rt runs one wave per tile for both sweeps (>= 4096 tiles) the chunked form saves the passes over the plane; in
    // between (the bench column at full size: 1563 tiles, two waves per tile) the separate kernels are as fast (profiles/r04_notes.md)
    const int nt64 = (int)((c.nnu + 63) / 64);
    if ((ctx->tune[15] & 3) != 2 && c.rtg.ud) return 0;
    const int nw = 4, R = 16 + c.nlob - 1;
    const size_t sh = ((size_t)2 * np * nw + (size_t)nw * R * 64) * sizeof(double);
    if (sh > lim) return 0;
    *shmem = sh; *nblk = (nt64 + nw - 1) / nw; *threads = nw * 64;
    return 2;
}

template <int NS>
void launch_flux_ns(int form, size_t shmem, int nblk, int threads, hipStream_t s, const RtParams &p, const double *nu, const double *wts,
                           int64_t nnu, const double *sigma, const double *muk, const double *P, const double *Tlev, const double *S,
                           const double *alb, double *tau, double *Mup, double *Mdn, double *partial, const FluxFuse &f, bool three_waves = true, bool scan_recompute = false)
{
    if (form == 3) {
        if constexpr (NS >= 2 && NS <= 8) {
            // a chunk of up to 5 layers (60 layers over 12 waves) keeps its transmissivities in registers between the sweeps and passes
            const int nw = threads / 64, per = (p.np - 1 + nw - 1) / nw;
            bool in_regs = false;
            if constexpr (NS <= 6) {   // (seven streams and up: the 5 x NS values no longer fit beside the rest at three waves per SIMD)
                if (per <= 5 && !scan_recompute) {
                    if (shmem > 65536) (void)hipFuncSetAttribute((const void *)k_flux_scan<NS, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                    CS_LAUNCH((k_flux_scan<NS, 5>), dim3(nblk), dim3(threads), shmem, s, p, nu, wts, nnu, sigma, muk, P, Tlev, S, alb, tau, Mup, Mdn, partial, f);
                    in_regs = true;
                }
            }
            if (!in_regs) {
                if (shmem > 65536) (void)hipFuncSetAttribute((const void *)k_flux_scan<NS, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
                CS_LAUNCH((k_flux_scan<NS, 0>), dim3(nblk), dim3(threads), shmem, s, p, nu, wts, nnu, sigma, muk, P, Tlev, S, alb, tau, Mup, Mdn, partial, f);
            }
        }
    } else if (three_waves) {
        if (shmem > 65536) (void)hipFuncSetAttribute((const void *)k_flux_chunk3<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        CS_LAUNCH((k_flux_chunk3<NS>), dim3(nblk), dim3(threads), shmem, s, p, nu, wts, nnu, sigma, muk, P, Tlev, S, alb, tau, Mup, Mdn, partial, f,
                  (int)((nnu + 63) / 64));
    } else {
        if (shmem > 65536) (void)hipFuncSetAttribute((const void *)k_flux_chunk<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        CS_LAUNCH((k_flux_chunk<NS>), dim3(nblk), dim3(threads), shmem, s, p, nu, wts, nnu, sigma, muk, P, Tlev, S, alb, tau, Mup, Mdn, partial, f,
                  (int)((nnu + 63) / 64));
    }
}

// includedlines(::Vector) line_shapes.jl:18-22 -> [g0, g1) ; strict = 0 keeps every line (scalar-nu method :12-16)
void included_range(const std::vector<double> &nul, double numin, double numax, double cut, bool strict, int64_t &g0,
                    int64_t &g1)
{
    const int64_t L = (int64_t)nul.size();
    g0 = 0;
    g1 = L;
    if (strict) {
        const double lo = numin - cut, hi = numax + cut;
        while (g0 < L && !(nul[g0] > lo)) g0++;
        while (g1 > g0 && !(nul[g1 - 1] < hi)) g1--;
    }
}

// per-tile union windows (a superset of every lane's |nu - nul| <= cut run; the kernel applies the exact test)
void tile_windows(const std::vector<double> &nul, int64_t g0, int64_t g1, const double *nu, int64_t nnu, double cut,
                  std::vector<int32_t> &J0, std::vector<int32_t> &J1, int64_t &pairs, int64_t &inrange)
{
    const int ntile = (int)((nnu + 255) / 256);
    J0.resize(ntile);
    J1.resize(ntile);
    auto b = nul.begin() + g0, e = nul.begin() + g1;
    for (int t = 0; t < ntile; t++) {
        const int64_t i0 = (int64_t)t * 256, i1 = std::min<int64_t>(nnu, i0 + 256) - 1;
        const double lo = nu[i0] - cut, hi = nu[i1] + cut;
        const double tol = 1e-9 * (std::fabs(hi) + cut + 1.0);
        J0[t] = (int32_t)(std::lower_bound(b, e, lo - tol) - nul.begin());
        J1[t] = (int32_t)(std::upper_bound(b, e, hi + tol) - nul.begin());
    }
    pairs = -1;  // counted on demand (cs_column_counts): O(nnu log L) on the host
    inrange = (std::upper_bound(b, e, nu[nnu - 1] + cut) - std::lower_bound(b, e, nu[0] - cut));
}

// The far kernel hands k_voigt_near the near-line ranges of every (nu, node) as 20-bit offsets into the tile's near zone and
// 12-bit counts.  Bound both for any state (widest Doppler width: upper end of the grid, TMAX, lightest isotopologue) and
// refuse tables too dense for the fields -- 4096 lines within a few Doppler widths is ~1e5 lines per cm^-1, far beyond HITEMP.
int check_near_density(const GasTable &G, double nu_hi, double span, double cut);
int check_near_density(const GasTable &G, const double *nu, int64_t nnu, double cut)
{
    double span = 0.0;
    for (int64_t i0 = 0; i0 < nnu; i0 += 64) span = std::max(span, nu[std::min<int64_t>(i0 + 63, nnu - 1)] - nu[i0]);
    return check_near_density(G, nu[nnu - 1], span, cut);
}
int check_near_density(const GasTable &G, double nu_hi, double span, double cut)   // span: widest 64-point tile of the grid
{
    const double amax = ((nu_hi + cut) / kC) * std::sqrt(2.0 * kRgas * kTmax / G.mu_min);
    const double dA = 100.0 * amax / kSqLn2 * (1.0 + 1e-6), r0 = std::sqrt(kSerS) * amax / kSqLn2 * 1.01;
    const std::vector<double> &v = G.h_nu;
    auto max_in = [&](double width) {
        int64_t best = 0;
        size_t a = 0;
        for (size_t b = 0; b < v.size(); b++) {
            while (v[b] - v[a] > width) a++;
            best = std::max<int64_t>(best, (int64_t)(b - a + 1));
        }
        return best;
    };
    const int64_t nzone = max_in(span + 2.0 * dA), npt = max_in(2.0 * r0);
    if (nzone >= (1 << 20) || npt >= (1 << 12))
        return fail(CS_EINVAL, "line table too dense for the near-line hand-off: %lld lines within %.3g cm^-1, %lld within %.3g cm^-1",
                    (long long)nzone, span + 2.0 * dA, (long long)npt, 2.0 * r0);
    return CS_OK;
}

// Doppler profile exp(-(dnu/alpha)^2) (line_shapes.jl:160): beyond sqrt(750) widths it is an exact zero in fp64 (exp(-745.2) is
// the smallest denormal), so the windows of the Doppler line sum only need the lines within that reach of a tile -- a bound for
// every state: the widest line is the one at the upper end of the grid, at TMAX, of the lightest isotopologue (line_shapes.jl:144)
double window_reach(int shape, const GasTable &G, double numax, double cut)
{
    if (shape != SH_DOPPLER) return cut;
    const double amax = ((numax + cut) / kC) * std::sqrt(2.0 * kRgas * kTmax / G.mu_min);
    return std::min(cut, std::sqrt(750.0) * amax * (1.0 + 1e-6));
}

int64_t count_pairs(const std::vector<double> &nul, const double *nu, int64_t nnu, double cut)
{
    int64_t pairs = 0;
    for (int64_t i = 0; i < nnu; i++)
        pairs += (std::upper_bound(nul.begin(), nul.end(), nu[i] + cut) - std::lower_bound(nul.begin(), nul.end(), nu[i] - cut));
    return pairs;
}

// per-64-point windows of the Voigt fast path: [W0,W1) superset window, [E0,E1) lines inside every lane's cut-off
// For span == 64 (the tiles of k_voigt_far) three more entries follow the nt windows: the block order of that kernel (tile_block).
// xc[0..8] = tile indices (multiples of 4) of eight contiguous stretches of the spectrum, one per XCD, when the line table is
// dense enough for its records to matter in that kernel's HBM traffic (32 B x lines against 16 B x wavenumbers per state:
// contiguous from L >= nnu / 8); xc[0] = -1 = plain block order otherwise.  Returns the number of tiles per XCD (grid size).
int wave_windows(const std::vector<double> &nul, int64_t g0, int64_t g1, const double *nu, int64_t nnu, double cut,
                 std::vector<WaveWin> &win, int span = 64)
{
    const int nt = (int)((nnu + span - 1) / span);
    win.resize(nt);
    auto b = nul.begin() + g0, e = nul.begin() + g1;
    for (int t = 0; t < nt; t++) {
        const int64_t i0 = (int64_t)t * span, i1 = std::min<int64_t>(nnu, i0 + span) - 1;
        const double vlo = nu[i0], vhi = nu[i1];
        const double tol = 1e-9 * (std::fabs(vhi) + cut + 1.0);
        WaveWin w;
        w.W0 = (int32_t)(std::lower_bound(b, e, vlo - cut - tol) - nul.begin());
        w.W1 = (int32_t)(std::upper_bound(b, e, vhi + cut + tol) - nul.begin());
        w.E0 = (int32_t)(std::lower_bound(b, e, vhi - cut + tol) - nul.begin());
        w.E1 = (int32_t)(std::upper_bound(b, e, vlo + cut - tol) - nul.begin());
        w.E0 = std::min(std::max(w.E0, w.W0), w.W1);
        w.E1 = std::min(std::max(w.E1, w.E0), w.W1);
        win[t] = w;
    }
    if (span != 64) return 0;
    int32_t xc[12] = {0};
    const int nt4 = (nt + 3) / 4 * 4;
    const int per = ((nt4 / 4 + 7) / 8) * 4;      // tiles per XCD, a multiple of 4
    const int64_t inrange = (std::upper_bound(b, e, nu[nnu - 1] + cut) - std::lower_bound(b, e, nu[0] - cut));
    for (int x = 0; x <= 8; x++) xc[x] = std::min(x * per, nt4);
    if (inrange * 8 < nnu) xc[0] = -1;
    win.resize(nt + 3);
    memcpy(&win[nt], xc, sizeof xc);
    return per;
}

// upper bound of gammalorentz (line_shapes.jl:255-257) over a gas's lines at every state
std::vector<double> gamma_bound(const GasTable &G, int K, const double *T, const double *P, const double *Pp)
{
    std::vector<double> g(K);
    for (int k = 0; k < K; k++) {
        const double r = kTref / T[k];
        const double f = std::max(std::pow(r, G.na_min), std::pow(r, G.na_max));
        g[k] = f * (std::fabs(G.ga_max * (P[k] - Pp[k])) + std::fabs(G.gs_max * Pp[k])) / kAtm * (1.0 + 1e-12);
    }
    return g;
}

// what k_gas_setup needs per state alone: ln(Tref/T) and Qref/Q(T) of every isotopologue of the table (line_shapes.jl:27-48)
void state_tables(const GasTable &G, int K, const double *T, std::vector<double> &lrt, std::vector<double> &qrefq)
{
    lrt.resize(K);
    qrefq.assign((size_t)K * G.niso, 0.0);
    for (int k = 0; k < K; k++) {
        lrt[k] = std::log(kTref / T[k]);
        for (int i = 0; i < G.niso; i++)
            if (G.h_ncheb[i] > 0) qrefq[(size_t)k * G.niso + i] = cheby_qrefq(T[k], G.h_ncheb[i], G.h_cheb.data() + (size_t)i * CS_CHEB_LD);
    }
}

// far wings by interpolation (k_cheb_nodes + k_cheb_apply), view handed to launch_gas; nlev = 0: off
struct Interp {
    int nlev = 0, nItot = 0, Kpad = 0, l0 = 0;
    int itv[CS_MAX_LEVEL], nI[CS_MAX_LEVEL], ioff[CS_MAX_LEVEL];
    const double *nodes = nullptr, *Cm[CS_MAX_LEVEL];
    const WaveWin *iwin[CS_MAX_LEVEL];
    IZone *iz = nullptr;
    double *F = nullptr;
    SepZone *sep = nullptr;   // NULL: every node sum on the vector unit
    EdgeZone *edge = nullptr; // NULL: all of the per-point sum on the vector unit
    bool sep_always = false;  // cs_set_matrix_cores(ctx, 2): also on grids too short to fill the chip with (interval, state group) blocks
    bool core = true;         // cs_set_matrix_cores(ctx, on | 4) switches the sub-tile treatment of the window core (k_voigt_sub) off
    double margin = kChebMargin;   // cs_set_tuning key 3 (per cent): distance of an interval's interpolated set, in half-widths
    int mx_min_states = 7;    // cs_set_tuning key 8
    int nfar[CS_MAX_LEVEL] = {};   // nodes for the far pieces of a level in k_cheb_nodes_mx (16, 32; 64 = as the near pieces)
    const double *Rc[CS_MAX_LEVEL] = {};   // ChebGrid::Rc
    int edge_phases = 1;           // cs_set_tuning key 14: k_voigt_edge_mx cuts a cut-off edge by the sub-tiles its lines reach (0: off)
    int nodes_split = 0;           // cs_set_tuning key 13: k_cheb_nodes with four waves per (interval, state)
    int cascade = 0;               // cs_set_tuning key 12: 0 = where it pays (cascade_pays), 1 = always, 2 = never
    const double *R = nullptr;     // the context's re-interpolation matrices (cs_ctx::reinterp); NULL: every piece on 64 nodes (cs_set_tuning key 11)
    int nsplit_levels = 1;    // cs_set_tuning key 6: interval sizes (largest first) whose node sums four waves share in k_cheb_nodes_mx
    bool small_mx = false;    // cs_set_tuning key 1: the matrix-core kernels on short grids too (their four-waves-per-item variants)
    bool mxzones_one_thread = false;   // cs_set_tuning key 15 | 16: k_mxzones instead of k_mxzones16
    const double *tnodes = nullptr, *tC = nullptr;   // ChebGrid::tnodes, tC where the grid allows (tile_nodes_ok) and cs_set_tuning key 23 = 0
    int far_split = 0;                 // cs_set_tuning key 22: waves per tile of k_voigt_far (1, 2, 4; 0 = by grid size)
    bool far_shared_full = false;      // cs_set_tuning key 17: far pieces of an item the four waves of a block share on all 64 nodes (A/B)
    int mxzones_merge = 0;             // cs_set_tuning key 21: the piece tables as blocks of k_gas_setup's launch (k_gas_setup_mx) -- 0 = on grids below 1024 tiles, 1 = never, 2 = always
    bool near_both = true;             // both tiers of the near-line pairs in one launch where a wave takes one tile (cs_set_tuning key 16 | 4: off)
    int near_prio = 0;                 // cs_set_tuning key 16: issue priority for k_voigt_sub / k_voigt_near (0 = from 512 tiles on, 1 = never, 2 = always)
    bool fuse_apply = false;  // the column's only interpolating group: k_voigt_edge_mx may carry the node sums to the grid itself
    bool near_memset = false; // cs_set_tuning key 19: the near-line plane cleared by a memset in front of k_voigt_sub (1) instead of written by it (0, default)
    double core4 = 0.0;       // the core takes the 4-term series where its radius is below core4 x the tile's span, else the 8-term one
                              // (0: always the 8-term one -- measured at C3 with 0.75 / 0.3 / 0: 2.61 / 2.55 / 2.52 ms)
};

static void interp_settings(const cs_ctx *ctx, Interp &itp)   // the cs_set_tuning keys an Interp view carries
{
    itp.small_mx = ctx->tune[1] != 0;
    itp.margin = ctx->tune[3] > 0 ? 0.01 * ctx->tune[3] : kChebMargin;
    itp.nsplit_levels = ctx->tune[6] > 0 ? ctx->tune[6] : 1;
    itp.mx_min_states = ctx->tune[8] > 0 ? ctx->tune[8] : 7;
    itp.R = ctx->tune[11] ? nullptr : ctx->reinterp.as<double>();
    itp.cascade = ctx->tune[12];
    itp.nodes_split = ctx->tune[13];
    itp.edge_phases = ctx->tune[14] ? 0 : 1;
    itp.mxzones_one_thread = (ctx->tune[15] & 16) != 0;
    itp.near_prio = ctx->tune[16] & 3;
    itp.near_both = (ctx->tune[16] & 4) == 0;
    itp.near_memset = ctx->tune[19] != 0;
    itp.mxzones_merge = ctx->tune[21];
    itp.far_split = ctx->tune[22];
    itp.far_shared_full = ctx->tune[17] != 0;
    if (ctx->tune[23]) itp.tnodes = itp.tC = nullptr;     // (cs_set_tuning key 23 = 1: every window-end line at the points, A/B)
}

// interval sizes worth using on this grid: an interval of width W leaves lines over (2 cut - 2.3 W) to interpolate
int choose_levels(double nu_lo, double nu_hi, int64_t nnu, double cut, int *itv, int szmin = 128, int szmax = 2048)
{
    int n = 0;
    if (nnu < 128) return 0;
    const double dnu = (nu_hi - nu_lo) / (double)(nnu - 1);
    for (int sz = 8192; sz >= 128 && n < CS_MAX_LEVEL; sz >>= 1)
        if (sz <= szmax && sz >= szmin && 2.3 * sz * dnu <= 1.5 * cut && sz / 2 <= nnu &&
            sz * dnu > 1e-8 * std::fabs(nu_hi))   // (nodes 1e-3 of an interval apart must stay distinct doubles)
            itv[n++] = sz;
    return n;
}
int choose_levels(const double *nu, int64_t nnu, double cut, int *itv, int szmin = 128, int szmax = 2048)
{
    return nnu < 1 ? 0 : choose_levels(nu[0], nu[nnu - 1], nnu, cut, itv, szmin, szmax);
}

// the grid part: nodes [nItot][64] and interpolation matrices [nI][64][itv] per level
int cheb_build_range(ChebGrid &g, double nu_lo, double nu_hi, const double *dnu, int64_t nnu, double cut, int szmin, int szmax, hipStream_t s);
int far_node_count(double dist, double h);
int cheb_build(const cs_ctx *ctx, ChebGrid &g, const double *h_nu, const double *dnu, int64_t nnu, double cut, hipStream_t s)
{
    const int rc = cheb_build_range(g, h_nu[0], h_nu[nnu - 1], dnu, nnu, cut, ctx->itp_min, ctx->itp_max, s);
    for (int l = 0; l < g.nlev; l++) {
        g.span[l] = 0.0;
        for (int64_t i0 = 0; i0 < nnu; i0 += g.itv[l]) g.span[l] = std::max(g.span[l], h_nu[std::min<int64_t>(i0 + g.itv[l] - 1, nnu - 1)] - h_nu[i0]);
    }
    // the window-end lines inside the cut-off of every point of a tile are at least cut-off - (smallest interval) - (tile) from it:
    // 16 nodes of the tile carry their sum to rounding where far_node_count says so (23 cm^-1 from a 1.6 cm^-1 tile: by far)
    g.tile_nodes_ok = false;
    if (g.nlev > 0) {
        double tspan = 0.0;
        for (int64_t i0 = 0; i0 < nnu; i0 += 64) tspan = std::max(tspan, h_nu[std::min<int64_t>(i0 + 63, nnu - 1)] - h_nu[i0]);
        const double dist = cut - g.span[g.nlev - 1] - tspan;
        g.tile_nodes_ok = tspan > 0.0 && dist > 0.0 && far_node_count(dist, 0.5 * tspan) == 16;
    }
    return rc;
}
// nodes for a set of lines that stays `dist` away from intervals of half-width h: the interpolation error falls like rho^-(n-1),
// rho = x0 + sqrt(x0^2 - 1), x0 = 1 + dist / h; 1e-18 asked for (64 nodes at the margin of 0.3: rho = 2.1, 5e-21)
int far_node_count(double dist, double h)
{
    if (!(dist > 0.0) || !(h > 0.0)) return CS_NC;
    const double x0 = 1.0 + dist / h, lr = std::log10(x0 + std::sqrt(x0 * x0 - 1.0));
    return 15.0 * lr >= 18.0 ? 16 : (31.0 * lr >= 18.0 ? 32 : CS_NC);
}
int cheb_build_range(ChebGrid &g, double nu_lo, double nu_hi, const double *dnu, int64_t nnu, double cut, int szmin, int szmax, hipStream_t s)
{
    g.nlev = choose_levels(nu_lo, nu_hi, nnu, cut, g.itv, szmin, szmax);
    g.nItot = 0;
    for (int l = 0; l < g.nlev; l++) {
        g.nI[l] = (int)((nnu + g.itv[l] - 1) / g.itv[l]);
        g.ioff[l] = g.nItot;
        g.nItot += g.nI[l];
    }
    if (g.nlev == 0) return CS_OK;
    HIPCHK(g.nodes.reserve((size_t)g.nItot * CS_NC * sizeof(double)));
    for (int l = 0; l < g.nlev; l++) {
        HIPCHK(g.Cm[l].reserve((size_t)g.nI[l] * CS_NC * g.itv[l] * sizeof(double)));
        CS_LAUNCH(k_cheb_setup, dim3(g.nI[l]), dim3(256), 0, s, dnu, nnu, g.itv[l], g.nI[l], CS_NC,
                           g.nodes.as<double>() + (size_t)g.ioff[l] * CS_NC, g.Cm[l].as<double>());
        HIPCHK(hipGetLastError());
    }
    for (int l = 1; l < g.nlev; l++) {
        int pshift = 0;
        for (int r = g.itv[l - 1] / g.itv[l]; r > 1; r >>= 1) pshift++;
        HIPCHK(g.Rc[l].reserve((size_t)g.nI[l] * CS_NC * CS_NC * sizeof(double)));
        CS_LAUNCH(k_cascade_setup, dim3(g.nI[l]), dim3(256), 0, s, g.nodes.as<double>(), g.ioff[l - 1], g.ioff[l], pshift, g.Rc[l].as<double>());
        HIPCHK(hipGetLastError());
    }
    {   // the tiles as intervals of their own with 16 nodes (k_voigt_edge_mx's node path)
        const int nt64 = (int)((nnu + 63) / 64);
        HIPCHK(g.tnodes.reserve((size_t)nt64 * 16 * sizeof(double)));
        HIPCHK(g.tC.reserve((size_t)nt64 * 16 * 64 * sizeof(double)));
        CS_LAUNCH(k_cheb_setup, dim3(nt64), dim3(256), 0, s, dnu, nnu, 64, nt64, 16, g.tnodes.as<double>(), g.tC.as<double>());
        HIPCHK(hipGetLastError());
    }
    return CS_OK;
}

int cheb_kpad(int K) { return (K + CS_KPAD - 1) / CS_KPAD * CS_KPAD; }

// the gas part: per-level interval windows (uploaded) and zone workspace for K states; F workspace of the grid
// First level worth using for a gas with `lambda` lines per cm^-1 (work per (nu, state) in line evaluations): a level costs
// ~4.4 evaluations in k_cheb_apply whatever the table, and saves (lines of its set) x (1/r_next - 1/r), r = interval size / 64.
int choose_l0(const ChebGrid &g, const double *nu, int64_t nnu, double cut, double lambda)
{
    const double dnu = (nu[nnu - 1] - nu[0]) / (double)std::max<int64_t>(nnu - 1, 1);
    int best = g.nlev;
    double bestc = 1e300;
    for (int l0 = 0; l0 <= g.nlev; l0++) {
        double c = 0.0;
        if (l0 == g.nlev) {
            c = 1.05 * 2.0 * cut * lambda;   // every pair evaluated directly
        } else {
            const double Wtop = g.itv[l0] * dnu;
            c += std::max(2.0 * cut - 2.3 * Wtop, 0.0) * lambda * 64.0 / g.itv[l0];
            for (int l = l0 + 1; l < g.nlev; l++) c += 2.3 * (g.itv[l - 1] - g.itv[l]) * dnu * lambda * 64.0 / g.itv[l];
            c += (2.3 * g.itv[g.nlev - 1] + 64.0) * dnu * lambda * 1.4;   // left to the per-point kernels
            c += 4.4 * (g.nlev - l0);
        }
        if (c < bestc) { bestc = c; best = l0; }
    }
    return best;
}

int gas_interp_build(const cs_ctx *ctx, GasInterp &gi, ChebGrid &g, const std::vector<double> &nul, int64_t g0, int64_t g1,
                     const double *nu, int64_t nnu, double cut, int K, hipStream_t s, bool own_F = true)
{
    int rc;
    gi.nlev = g.nlev;
    gi.l0 = 0;
    if (g.nlev == 0) return CS_OK;
    {
        const double span = nu[nnu - 1] - nu[0] + 2.0 * cut;
        const auto b = std::lower_bound(nul.begin() + g0, nul.begin() + g1, nu[0] - cut);
        const auto e = std::upper_bound(nul.begin() + g0, nul.begin() + g1, nu[nnu - 1] + cut);
        gi.l0 = choose_l0(g, nu, nnu, cut, (double)(e - b) / span);
        if (ctx->itp_first >= 0) gi.l0 = std::min(ctx->itp_first, g.nlev);
        if (gi.l0 >= g.nlev) { gi.nlev = 0; gi.l0 = 0; return CS_OK; }   // too few lines: every pair directly
    }
    // far pieces of a level (the lines beyond its parent's set): at least cut-off minus the parent's width from the interval
    // The carry from those nodes to the interval's 64 is a FIXED matrix in the interval's own coordinate, while both node sets are
    // doubles near nu: a sample taken half an ulp(nu) from its ideal place is off by 2 (ulp / 2) / distance of a 1/dnu^2 wing, and
    // nothing downstream knows (the 64-node sums go through matrices built from the stored nodes themselves, and the tile nodes of
    // k_voigt_edge_mx likewise).  Fewer nodes only where ulp / distance <= 8e-14 (cut-off 25 cm^-1 at 2500 cm^-1: 2.4e-14 / 3.7e-14;
    // a cut-off of 1 cm^-1 there would give 9e-13 -- tests/test_gpu_fuzz.py::test_interp_fuzz[8] once the short grids took this path)
    const double ulp_hi = std::nextafter(std::fabs(nu[nnu - 1]), INFINITY) - std::fabs(nu[nnu - 1]);
    for (int l = 0; l < g.nlev; l++) {
        gi.nfar[l] = (l > gi.l0 && g.span[l - 1] > 0.0 && g.span[l] > 0.0) ? far_node_count(cut - g.span[l - 1], 0.5 * g.span[l]) : CS_NC;
        if (gi.nfar[l] < CS_NC && !(ulp_hi <= 8e-14 * (cut - g.span[l - 1]))) gi.nfar[l] = CS_NC;
    }
    for (int l = 0; l < g.nlev; l++) {
        std::vector<WaveWin> iwin;
        wave_windows(nul, g0, g1, nu, nnu, cut, iwin, g.itv[l]);
        if ((rc = upload(gi.iwin[l], iwin.data(), iwin.size(), s))) return rc;
        HIPCHK(hipStreamSynchronize(s));   // iwin is a local
    }
    HIPCHK(gi.iz.reserve((size_t)K * g.nItot * sizeof(IZone)));
    HIPCHK(gi.sep.reserve((size_t)((K + 15) / 16) * g.nItot * sizeof(SepZone)));
    HIPCHK(gi.edge.reserve((size_t)((K + 15) / 16) * (size_t)((nnu + 63) / 64) * sizeof(EdgeZone)));
    if (own_F && gi.F.bytes < (size_t)g.nItot * CS_NC * cheb_kpad(K) * sizeof(double)) {
        HIPCHK(gi.F.reserve((size_t)g.nItot * CS_NC * cheb_kpad(K) * sizeof(double)));
        HIPCHK(hipMemsetAsync(gi.F.p, 0, gi.F.bytes, s));   // padding states stay finite
    }
    return CS_OK;
}

Interp interp_view(const ChebGrid &g, const GasInterp &gi, int K, IZone *iz_override = nullptr)
{
    Interp v;
    v.nlev = gi.nlev;
    v.l0 = gi.l0;
    v.nItot = g.nItot;
    v.Kpad = cheb_kpad(K);
    v.nodes = g.nodes.as<double>();
    v.iz = iz_override ? iz_override : gi.iz.as<IZone>();
    v.F = gi.F.as<double>();
    v.sep = gi.sep.as<SepZone>();
    v.edge = gi.edge.as<EdgeZone>();
    if (g.tile_nodes_ok) { v.tnodes = g.tnodes.as<double>(); v.tC = g.tC.as<double>(); }
    for (int l = 0; l < gi.nlev; l++) {
        v.itv[l] = g.itv[l]; v.nI[l] = g.nI[l]; v.ioff[l] = g.ioff[l];
        v.nfar[l] = gi.nfar[l];
        v.Cm[l] = g.Cm[l].as<double>();
        v.Rc[l] = g.Rc[l].as<double>();
        v.iwin[l] = gi.iwin[l].as<WaveWin>();
    }
    return v;
}

void launch_apply(hipStream_t s, const ChebApply &A, int Kpad, int64_t nnu, int kn, double base, const double *extra, double *sigma,
                  int accumulate, bool varnc = false);
// the node sums of ONE F (levels l0 ..) to the grid: level by level into the next smaller one, then the smallest (k_cheb_cascade), or
// every level by itself
// (measured: five levels, BASELINE configs[4]: apply 1.31 -> 1.12 ms; three levels, the bench column: 0.092 -> 0.100 ms, a 1/8 shard
//  of it 0.029 -> 0.037 ms -- F goes through memory once more per level, and each level is one more launch)
static bool cascade_pays(int nlevels_in_use) { return nlevels_in_use >= 4; }
void launch_apply_cascade(hipStream_t s, ChebApply A, const double *const *Rc, const int *itv, const int *nI, int mode, int Kpad, int64_t nnu,
                          int kn, double base, const double *extra, double *sigma, int accumulate, ChebApply *carry = nullptr /* != NULL: the
                          cascade only; *carry = what is still to be carried to the grid (k_flux does that) */)
{
    const int l0 = A.l0[0];
    const bool on = A.ngas == 1 && A.nlev - l0 >= 2 && mode != 2 && (mode == 1 || cascade_pays(A.nlev - l0));
    if (on) {
        const int nst = cheb_kpad(kn) / 16;
        double *F = const_cast<double *>(A.F[0]);
        for (int l = l0 + 1; l < A.nlev; l++) {
            int pshift = 0;
            for (int r = itv[l - 1] / itv[l]; r > 1; r >>= 1) pshift++;
            CS_LAUNCH(k_cheb_cascade, dim3((unsigned)(((int64_t)nI[l] * nst + 3) / 4)), dim3(256), 0, s, Rc[l], F, A.ioff[l - 1], A.ioff[l], pshift,
                      nI[l], Kpad, nst);
        }
        A.l0[0] = A.nlev - 1;
    }
    if (carry) { *carry = A; return; }
    launch_apply(s, A, Kpad, nnu, kn, base, extra, sigma, accumulate);
}
void launch_apply(hipStream_t s, const ChebApply &A, int Kpad, int64_t nnu, int kn, double base, const double *extra, double *sigma,
                  int accumulate, bool varnc)
{
    const int nt64 = (int)((nnu + 63) / 64);
    const int kp = cheb_kpad(kn);                // sub-tiles actually in use (Kpad is the row pitch of F)
    const int nst = kp / 16;
    const unsigned tb8 = (unsigned)(((nt64 + 3) / 4 + 7) / 8 * 8);
    (void)Kpad;
#ifndef CS_APPLY_NSUB
#define CS_APPLY_NSUB 2
#endif
    const bool big = (int64_t)nt64 * ((nst + CS_APPLY_NSUB - 1) / CS_APPLY_NSUB) >= 2048;   // enough (tile, state chunk) waves to fill 1024 SIMDs twice
    const dim3 gridb(tb8 * (unsigned)((nst + CS_APPLY_NSUB - 1) / CS_APPLY_NSUB)), grids(tb8 * (unsigned)nst);
    if (varnc) {
        if (big) CS_LAUNCH((k_cheb_apply_mfma<CS_APPLY_NSUB, true>), gridb, dim3(256), 0, s, A, Kpad, nnu, nt64, kn, base, extra, sigma, accumulate);
        else CS_LAUNCH((k_cheb_apply_mfma<1, true>), grids, dim3(256), 0, s, A, Kpad, nnu, nt64, kn, base, extra, sigma, accumulate);
    } else {
        if (big) CS_LAUNCH((k_cheb_apply_mfma<CS_APPLY_NSUB, false>), gridb, dim3(256), 0, s, A, Kpad, nnu, nt64, kn, base, extra, sigma, accumulate);
        else CS_LAUNCH((k_cheb_apply_mfma<1, false>), grids, dim3(256), 0, s, A, Kpad, nnu, nt64, kn, base, extra, sigma, accumulate);
    }
}

// PHCO2 fast path preconditions + workspace.  Returns false -> generic kernel.
bool phco2_fast_ok(const GasTable &G, int64_t nnu, double cut, int kn, PhScratch *ph)
{
    if (cut < 130.0 || nnu < 2 || !(ph->nu_hi > ph->nu_lo)) return false;
    // widest near zone of any state must stay inside the chi = 1 core (|dnu| < 3 cm^-1) of its tile
    const double amax = ((ph->nu_hi + cut) / kC) * std::sqrt(2.0 * kRgas * kTmax / G.mu_min);
    if (100.0 * amax / kSqLn2 * (1.0 + 1e-6) > 2.9) return false;
    const int nt64 = (int)((nnu + 63) / 64);
    if ((ph->nu_hi - ph->nu_lo) / (double)(nnu - 1) * 64.0 > 5.0) return false;   // (mean tile span: wider tiles would not be faster)
    if (!(ph->max_span <= 27.0)) return false;   // k_phco2's boundary sets hold one region boundary each: no tile wider than 30 - 3 cm^-1
    // the tabulated line factors exp(+-b_r (nul - nu_c)) must stay in range: b_r <= 0.0888
    if (0.0888 * (0.5 * (ph->nu_hi - ph->nu_lo) + cut) > 600.0) return false;
    if (ph->fac.reserve((size_t)6 * kn * G.L * sizeof(double)) != hipSuccess) return false;
    if (ph->win.reserve((size_t)nt64 * sizeof(PhWin)) != hipSuccess) return false;
    ph->nu_c = 0.5 * (ph->nu_lo + ph->nu_hi);
    return true;
}
// the grid of the coming launch_gas calls (host copy `nu`), and the context's interpolation settings
static std::atomic<uint64_t> g_grid_counter{0};
// interpolation levels of the PHCO2 far wings on that grid; false: none (k_phco2 sums every pair per point)
static int ph_plan(PhScratch *ph, int64_t nnu, double cut)
{
    PhScratch::Grid &g = ph->grid;
    memset(&g.lv, 0, sizeof g.lv);
    memset(&g.vl, 0, sizeof g.vl);
    for (int r = 0; r < 3; r++) { g.fine.off[r] = -1; g.fine.shift[r] = 0; }
    g.nnodes = g.nslots = 0;
    if (nnu < 128) return 0;
    const double dnu = (ph->nu_hi - ph->nu_lo) / (double)(nnu - 1);
    // (cs_set_interp_plan's limits of 128 and 2048 points are the Voigt path's; left there, the wide PHCO2 window also takes 4096 and
    //  8192 points; key 10 bit 1: also the tiles themselves as 64-point intervals with 16 or 32 nodes -- measured a tie at C3 size: 1.2 ms
    //  more in k_phco2_nodes, whose smallest waves are a chain of short batches, for 1.3 ms less in k_phco2)
    const int szmax = ph->itp_max >= 2048 ? 8192 : ph->itp_max, szmin = (ph->itp_min <= 128 && (ph->force64 & 2)) ? 64 : ph->itp_min;
    const double Dn[3] = {3.0, 30.0, 120.0}, Df[3] = {30.0, 120.0, cut};
    int need[8][3];   // per size and region: node count, 0 = not carried
    int nreal = 0, real_of[8];
    for (int i = 0; i < 8; i++) {
        const int sz = 8192 >> i;
        real_of[i] = -1;
        if (!(sz <= szmax && sz >= szmin && 2.3 * sz * dnu <= 1.5 * cut && sz / 2 <= nnu && sz * dnu > 1e-8 * std::fabs(ph->nu_hi))) continue;
        const double span = ph->lev_span[i], h = 0.5 * span;
        bool any = false;
        for (int r = 0; r < 3; r++) {
            const double dn = std::max(Dn[r], ph->margin * h);
            need[i][r] = 0;
            if (!(Df[r] - dn - sz * dnu >= 0.05 * (Df[r] - Dn[r]))) continue;   // (next to) nothing to hold at this size
            const double x0 = 1.0 + std::max(Dn[r] / h, ph->margin), lr = std::log10(x0 + std::sqrt(x0 * x0 - 1.0));
            need[i][r] = (ph->force64 & 1) ? 64 : (15.0 * lr >= 18.0 ? 16 : (31.0 * lr >= 18.0 ? 32 : 64));
            any = true;
        }
        if (!any) continue;
        real_of[i] = nreal;
        g.lv.itv[nreal] = sz;
        g.lv.nI[nreal] = (int)((nnu + sz - 1) / sz);
        g.lv.ioff[nreal] = g.lv.nItot;
        g.lv.nItot += g.lv.nI[nreal];
        nreal++;
    }
    g.lv.nlev = nreal;
    if (nreal == 0) return 0;
    // virtual levels; more than CS_MAX_ALEVEL of them: the regions of a size join its larger node count
    for (;;) {
        int nv = 0;
        for (int i = 0; i < 8; i++) {
            if (real_of[i] < 0) continue;
            for (int nc = 64; nc >= 16; nc >>= 1) {
                int mask = 0;
                for (int r = 0; r < 3; r++) if (need[i][r] == nc) mask |= 1 << r;
                if (mask) nv++;
            }
        }
        if (nv <= CS_MAX_ALEVEL) break;
        bool changed = false;
        for (int i = 7; i >= 0 && !changed; i--)
            for (int r = 0; r < 3 && !changed; r++)
                if (real_of[i] >= 0 && need[i][r] > 0 && need[i][r] < 64) { need[i][r] *= 2; changed = true; }
        if (!changed) break;
    }
    int nv = 0, last[3] = {-1, -1, -1};   // last[r]: the size above that carries region r
    for (int i = 0; i < 8; i++) {
        if (real_of[i] < 0) continue;
        for (int nc = 64; nc >= 16; nc >>= 1) {
            int mask = 0;
            for (int r = 0; r < 3; r++) if (need[i][r] == nc) mask |= 1 << r;
            if (!mask || nv >= CS_MAX_ALEVEL) continue;
            g.vl.rl[nv] = real_of[i]; g.vl.nc[nv] = nc; g.vl.rmask[nv] = mask;
            g.vl.noff[nv] = g.nnodes; g.vl.boff[nv] = g.nslots;
            for (int r = 0; r < 3; r++) g.vl.par[nv][r] = last[r];
            g.nnodes += g.lv.nI[real_of[i]] * nc;
            g.nslots += g.lv.nI[real_of[i]];
            nv++;
        }
        for (int r = 0; r < 3; r++)
            if (need[i][r] > 0) {
                last[r] = real_of[i];
                g.fine.off[r] = g.lv.ioff[real_of[i]];
                g.fine.shift[r] = 0;
                for (int x = (8192 >> i) / 64; x > 1; x >>= 1) g.fine.shift[r]++;
            }
    }
    g.vl.nv = nv;
    g.vl.boff[nv] = g.nslots;
    return nv;
}
bool ph_interp_ready(PhScratch *ph, const double *dnu, int64_t nnu, double cut, int kn, hipStream_t s)
{
    if (!ph->itp_on) return false;
    PhScratch::Grid &g = ph->grid;
    if (!(ph->built_id == ph->grid_id && ph->built_nnu == nnu && ph->built_cut == cut && ph->built_min == ph->itp_min &&
          ph->built_max == ph->itp_max && ph->built_margin == ph->margin && ph->built_force == ph->force64)) {
        ph->built_id = 0;
        const int nv = ph_plan(ph, nnu, cut);
        if (nv > 0) {
            if (g.nodes.reserve((size_t)g.nnodes * sizeof(double)) != hipSuccess) return false;
            for (int v = 0; v < nv; v++) {
                const int rl = g.vl.rl[v], nc = g.vl.nc[v];
                if (g.Cm[v].reserve((size_t)g.lv.nI[rl] * nc * g.lv.itv[rl] * sizeof(double)) != hipSuccess) return false;
                CS_LAUNCH(k_cheb_setup, dim3(g.lv.nI[rl]), dim3(256), 0, s, dnu, nnu, g.lv.itv[rl], g.lv.nI[rl], nc,
                          g.nodes.as<double>() + g.vl.noff[v], g.Cm[v].as<double>());
            }
            if (hipGetLastError() != hipSuccess) return false;
        }
        ph->built_id = ph->grid_id; ph->built_nnu = nnu; ph->built_cut = cut; ph->built_min = ph->itp_min; ph->built_max = ph->itp_max;
        ph->built_margin = ph->margin; ph->built_force = ph->force64;
    }
    if (g.vl.nv == 0) return false;
    if (ph->piw.reserve((size_t)g.lv.nItot * sizeof(PhIWin)) != hipSuccess) return false;
    const size_t fb = (size_t)g.nnodes * cheb_kpad(kn) * sizeof(double);
    if (ph->F.bytes < fb) {
        if (ph->F.reserve(fb) != hipSuccess) return false;
        if (hipMemsetAsync(ph->F.p, 0, ph->F.bytes, s) != hipSuccess) return false;   // padding states stay finite
    }
    return true;
}

// points per sub-tile of k_voigt_sub (16: 0.34 ms at C3 with every core on the 8-term series; 8: 0.22; 4: 0.22 -- what is left is not
// the evaluations)
constexpr int CS_SUBW = 8;
// per-point matrix pieces only on tables with at least this many lines per tile in range (C5: the synthetic O3 table at 6.4 per
// tile gains 0.12 ms with them, the HITRAN fixtures at 0.4-0.7 lose)
constexpr int CS_EDGE_DENS = 4;

// matrix-core node sums (k_cheb_nodes_mx): fp64 Voigt only, and by default only where there are enough (interval, state group)
// blocks to fill the chip -- on a short grid (a nu-shard) the one-state-per-wave vector kernel has the shorter critical path
// (1/8 of C3: 0.17 vs 0.25 ms)
static bool mx_big(int nblocks, int kn, int min_blocks) { return (int64_t)nblocks * ((kn + 15) / 16) >= min_blocks; }
// `small`: cs_set_tuning key 1 -- short grids too, through the variants that share one (interval | tile, group) between the four
// waves of a block (k_cheb_nodes_mx with every level split, k_voigt_edge_mx<4>)
static bool sep_in_use(bool have_sep, bool always, int nblocks_intervals, int kn, bool lor, bool mixed, bool small = false)
{
    (void)mixed;   // (the mixed-precision variant keeps the matrix-core pieces in fp64: only the vector bodies beyond far_s go to fp32)
    return have_sep && !lor && (always || small || mx_big(nblocks_intervals, kn, 2048));
}
// the per-point pieces (k_voigt_edge_mx: one wave per (tile, state group), no reduction) pay on shorter grids -- 1/4 of C3 (1564
// waves): far 0.42 -> 0.36 ms; 1/8: 0.261 -> 0.244 ms, which the extra zone launch eats -- but only on tables dense enough to give
// a wave more than a few steps (C5's HITRAN fixtures: far 3.94 -> 4.01 ms with them; its synthetic O3 table: step 10.63 -> 10.51)
static bool edge_in_use(bool have_edge, bool always, int ntiles, int kn, bool lor, bool mixed, int64_t lines_in_range, bool small = false)
{
    (void)mixed;
    return have_edge && !lor && (always || ((small || mx_big(ntiles, kn, 1024)) && lines_in_range >= (int64_t)ntiles * CS_EDGE_DENS));
}

// Short grids (a nu-shard): the kernels no longer fill the chip, a step is a chain of launch tails -- and the node sums (F) and the
// per-point kernels (sigma) of a group are independent until the interpolation carries F to the grid.  With a Fork the node kernels
// go to a side stream behind an event recorded after k_gas_setup / k_mxzones; `pending` says the main stream has not yet waited
// for them (it must before anything reads F or overwrites the records).
struct Fork {
    bool use_nodes, use_near;   // which of the two side streams this step uses
    hipStream_t s2; hipEvent_t ev_fork, ev_join; bool pending;
    // the same for the near-line kernels: they need the hand-off words of k_voigt_far / k_voigt_sub and nothing of k_voigt_edge_mx --
    // a gather-bound kernel beside a matrix-core one -- but both add to sigma, so the near-line pairs go to a plane of their own
    // (sigma2: cleared at the start of the step's first Voigt group, read together with sigma by k_rt, folded in by k_fold where sigma is
    // the result).  k_voigt_sub goes there too: it needs the zones only, not k_voigt_far, so it runs beside it.
    hipStream_t s3; hipEvent_t ev_fork3, ev_join3, ev_far3; bool pending3;
    double *sigma2; bool zeroed, live;
};
static void fork_join(Fork *f, hipStream_t s, bool nodes = true, bool near = true)
{
    if (f && nodes && f->pending) { (void)hipStreamWaitEvent(s, f->ev_join, 0); f->pending = false; }
    if (f && near && f->pending3) { (void)hipStreamWaitEvent(s, f->ev_join3, 0); f->pending3 = false; }
}

// K1 + K2 for one gas on `s`: parameters for `kn` states, then the line sum into sigma ([kn][nnu])
void launch_gas(hipStream_t s, int shape, const GasTable &G, int64_t jrange0, int64_t jrange1, int kn, const double *Tk, const double *Pk, const double *Ppk,
                const double *scale, int mstride /* members of a merged table: element (m, k) of Ppk / scale at m * mstride + k */,
                const double *lrt, const double *qrefq /* state_tables(): [kn], [kn][niso] */, LineHot *hot, LineCold *cold, const double *dnu, int64_t nnu, int ntile256,
                const int32_t *J0, const int32_t *J1, const WaveWin *win, int xtiles, Zone *zones, int2 *ranges, const double *gbound, double cut, double base,
                const double *extra, double *sigma, int accumulate, hipEvent_t *evg,   // NULL or 6 events: after K1 (+ zones), nodes (vector unit), nodes (matrix cores), far (vector unit), sub-tile cores, far (matrix cores)
                LineF32 *hot32 = nullptr, double far_s = 1e6, Interp itp = Interp(), ChebApply *defer = nullptr, PhScratch *ph = nullptr,
                Fork *fork = nullptr, bool records_ready = false /* hot / cold already hold this gas at these states: zones and sums only */)
{
    fork_join(fork, s);   // (an earlier group's node kernels may still read the records this launch overwrites)
    // only the lines some window can reach (windows are sorted: first tile's start .. last tile's end)
    const int64_t jlo = jrange0, jhi = std::max(jrange1, jrange0);
    PrepArgs pa;
    pa.shape = shape; pa.K = kn; pa.g = G.dev(); pa.jlo = jlo; pa.jhi = jhi;
    pa.Tk = Tk; pa.Pk = Pk; pa.Ppk = Ppk; pa.scale = scale; pa.mstride = mstride; pa.lrt = lrt; pa.qrefq = qrefq; pa.niso = G.niso; pa.hot = hot; pa.cold = cold; pa.hot32 = shape == SH_VOIGT ? hot32 : nullptr;
    pa.phfac = nullptr;
    pa.nu_c = 0.0;
    const unsigned nb_prep = records_ready ? 0u : (unsigned)((jhi - jlo + 255) / 256) * (unsigned)((kn + CS_PREP_KC - 1) / CS_PREP_KC);   // (line block, chunk of states)
    const bool lor = shape == SH_LORENTZ;   // lorentz! runs on the same far-wing machinery with its own (exact) body
    if (shape == SH_VOIGT || lor) {
        if (lor) hot32 = nullptr;           // (no fp32 variant of the Lorentz body)
        const int nt64 = (int)((nnu + 63) / 64);
        // wave priority of the near-line stream's kernels (wave_prio): long grids only
        const int near_prio = (itp.near_prio == 2 || (itp.near_prio == 0 && nt64 >= 512)) ? 3 : 0;
        ZoneArgs za;
        za.nu = dnu; za.nul = G.nu.as<double>(); za.Tk = Tk; za.gbound = gbound; za.win = win; za.zones = zones; za.nnu = nnu;
        za.lorentz = lor ? 1 : 0;
        za.ntile = nt64; za.K = kn; za.mu_min = G.mu_min; za.mu_max = G.mu_max; za.cut = cut; za.far_s = far_s;
        za.margin = itp.margin;
        const unsigned nb_zones = (unsigned)(((int64_t)nt64 * kn + 255) / 256);
        IzParams P;
        memset(&P, 0, sizeof P);
        unsigned nb_iz = 0;
        const IZone *iz = nullptr;
        int ishift = 0;
        bool forked_here = false;   // ev_fork was recorded on the main stream after the zone launches: the near-line side stream can wait on it too
        bool use_edge = false;   // window ends of the per-point sum on the matrix cores (k_voigt_edge_mx; with the far wings interpolated only)
        bool fuse = false;       // ... which then also applies the interpolated wings (no k_cheb_apply launch for this group)
        ChebApply Afuse;
        memset(&Afuse, 0, sizeof Afuse);
        if (itp.nlev > 0) {
            P.nlev = itp.nlev;
            P.nItot = itp.nItot;
            P.l0 = itp.l0;
            for (int l = 0; l < itp.nlev; l++) { P.itv[l] = itp.itv[l]; P.nI[l] = itp.nI[l]; P.ioff[l] = itp.ioff[l]; P.iwin[l] = itp.iwin[l]; }
            nb_iz = (unsigned)(((int64_t)(itp.nItot - itp.ioff[itp.l0]) * kn + 255) / 256);
        }
        // what the matrix cores take of the interpolated sets and of the window ends: piece tables per (interval | tile, state group).  They
        // need the zones -- which their sixteen-lanes-per-item form computes itself, as blocks of the SAME launch (k_gas_setup_mx); the
        // one-thread-per-item form (from ~50 000 items on: BASELINE configs[4]) reads them, a launch of its own behind k_gas_setup
        const int q0s = itp.nlev > 0 ? itp.ioff[itp.l0] : 0, ngrp_s = (kn + 15) / 16;
        const bool use_sep_s = itp.nlev > 0 && sep_in_use(itp.sep != nullptr, itp.sep_always, itp.nItot - q0s, kn, lor, hot32 != nullptr, itp.small_mx);
        if (itp.nlev > 0) use_edge = edge_in_use(itp.edge != nullptr, itp.sep_always, nt64, kn, lor, hot32 != nullptr, jhi - jlo, itp.small_mx);
        SepArgs sa;
        EdgeArgs ea;
        memset(&sa, 0, sizeof sa);
        memset(&ea, 0, sizeof ea);
        bool mx_one_thread = false, mx_merged = false;
        if (use_sep_s || use_edge) {
            sa.nodes = itp.nodes; sa.nul = G.nu.as<double>(); sa.gbound = gbound; sa.Tk = Tk; sa.iz = itp.iz; sa.out = itp.sep;
            sa.nItot = itp.nItot; sa.q0 = q0s; sa.K = kn; sa.ngrp = ngrp_s; sa.mu_min = G.mu_min; sa.cut = cut; sa.min_states = itp.mx_min_states;
            ea.nu = dnu; ea.nul = G.nu.as<double>(); ea.gbound = gbound; ea.Tk = Tk; ea.win = win; ea.zones = zones;
            ea.iz = itp.iz + itp.ioff[itp.nlev - 1]; ea.out = itp.edge; ea.nnu = nnu; ea.ntile = nt64; ea.K = kn; ea.ngrp = ngrp_s;
            ea.nI = itp.nItot; ea.ishift = 0;
            for (int r = itp.itv[itp.nlev - 1] / 64; r > 1; r >>= 1) ea.ishift++;
            ea.mu_min = G.mu_min; ea.cut = cut;
            ea.core = (use_edge && itp.core) ? 1 : 0;
            ea.core4 = itp.core4;
            // sixteen lanes per item shorten the chain where the items are few (a nu-shard: 21 -> 8 us; the bench column 27 -> 9); from
            // ~50 000 items on one thread per item has parallelism enough and sixteen times fewer threads (BASELINE configs[4]: 0.256 vs 0.276 ms)
            const int64_t nitems = (use_sep_s ? (int64_t)(itp.nItot - q0s) * ngrp_s : 0) + (use_edge ? (int64_t)nt64 * ngrp_s : 0);
            mx_one_thread = itp.mxzones_one_thread || nitems > 50000;   // (cs_set_tuning key 15 | 16: always)
            // merged on short grids, where the head of the step is a chain of launch tails (1/8 of the bench column: 0.347 -> 0.340 ms);
            // at full size the second set of searches beside 300 MB of record stores costs more than the launch it saves (1.900 -> 1.908)
            // (cs_set_tuning key 21: 1 = never, 2 = always, A/B)
            mx_merged = !mx_one_thread && itp.mxzones_merge != 1 && (itp.mxzones_merge == 2 || nt64 < 1024);
        }
        if (mx_merged) {
            const unsigned nb_sep = use_sep_s ? (unsigned)(((int64_t)(itp.nItot - q0s) * ngrp_s + 15) / 16) : 0u;
            const unsigned nb_edge = use_edge ? (unsigned)(((int64_t)nt64 * ngrp_s + 15) / 16) : 0u;
            CS_LAUNCH(k_gas_setup_mx, dim3(nb_prep + nb_zones + nb_iz + nb_sep + nb_edge), dim3(256), 0, s, nb_prep, nb_zones, nb_iz, nb_sep, pa, za, P, itp.iz, sa, ea);
        } else {
            CS_LAUNCH(k_gas_setup, dim3(nb_prep + nb_zones + nb_iz), dim3(256), 0, s, nb_prep, nb_zones, pa, za, P, itp.iz);
        }
        if (evg && itp.nlev == 0) (void)hipEventRecord(evg[0], s);
        if (itp.nlev > 0) {   // sigma = base + extra + interpolated far wings; the per-point kernels add the rest
            const int q0 = itp.ioff[itp.l0];
            // deferred apply: the gases of a column add their node sums into ONE F (levels an earlier gas has written accumulate)
            const int q_acc = (defer && defer->ngas > 0 && defer->l0[0] < itp.nlev) ? itp.ioff[defer->l0[0]] : itp.nItot;
            // short grids: four waves per (interval, state) (cs_set_tuning key 13: 0 = below 16384 waves, 1 = always, 2 = never)
            const bool nsplit4 = itp.nodes_split == 1 || (itp.nodes_split == 0 && (int64_t)(itp.nItot - q0) * kn < 16384);
            const dim3 gridn(nsplit4 ? (unsigned)kn * (unsigned)(itp.nItot - q0) : (unsigned)((kn + 3) / 4) * (unsigned)(itp.nItot - q0));
            const int ngrp = ngrp_s;
            const bool use_sep = use_sep_s;
            if ((use_sep || use_edge) && !mx_merged) {
                if (mx_one_thread) {
                    const unsigned nb_sep = use_sep ? (unsigned)(((int64_t)(itp.nItot - q0) * ngrp + 255) / 256) : 0u;
                    const unsigned nb_edge = use_edge ? (unsigned)(((int64_t)nt64 * ngrp + 255) / 256) : 0u;
                    CS_LAUNCH(k_mxzones, dim3(nb_sep + nb_edge), dim3(256), 0, s, nb_sep, sa, ea);
                } else {                        // sixteen lanes per item
                    const unsigned nb_sep = use_sep ? (unsigned)(((int64_t)(itp.nItot - q0) * ngrp + 15) / 16) : 0u;
                    const unsigned nb_edge = use_edge ? (unsigned)(((int64_t)nt64 * ngrp + 15) / 16) : 0u;
                    CS_LAUNCH(k_mxzones16, dim3(nb_sep + nb_edge), dim3(256), 0, s, nb_sep, sa, ea);
                }
            }
            if (evg) (void)hipEventRecord(evg[0], s);
            const SepZone *sepz = use_sep ? itp.sep : nullptr;
            hipStream_t sm = s;   // main stream
            if (fork && fork->use_nodes && defer && !evg) {
                (void)hipEventRecord(fork->ev_fork, s);
                (void)hipStreamWaitEvent(fork->s2, fork->ev_fork, 0);
                forked_here = true;
                s = fork->s2;     // the two node kernels below run beside what follows them on the main stream
            }
#define NODES_LAUNCH(M, L_, M4, L4, S4) do { if (nsplit4) CS_LAUNCH((k_cheb_nodes<M4, L4, S4>), gridn, dim3(256), 0, s, itp.nodes, G.L, hot, hot32, G.nu.as<double>(), itp.iz, \
                                   itp.nItot, q0, q_acc, kn, itp.Kpad, cut, itp.F, sepz); \
            else CS_LAUNCH((k_cheb_nodes<M, L_>), gridn, dim3(256), 0, s, itp.nodes, G.L, hot, hot32, G.nu.as<double>(), itp.iz, \
                                   itp.nItot, q0, q_acc, kn, itp.Kpad, cut, itp.F, sepz); } while (0)
            if (lor)
                NODES_LAUNCH(false, true, false, true, 4);
            else if (hot32)
                NODES_LAUNCH(true, false, true, false, 4);
            else
                NODES_LAUNCH(false, false, false, false, 4);
            if (evg) (void)hipEventRecord(evg[1], s);
            if (use_sep) {
                const int nq = itp.nItot - q0;
                // the largest interval sizes in use are shared by the four waves of a block -- itp.nsplit_levels of them (all sizes on a
                // grid too short to fill the chip with one (interval, group) per wave)
                int nsplit = 0;
                for (int l = itp.l0; l < std::min(itp.nlev, itp.l0 + itp.nsplit_levels); l++) nsplit += itp.nI[l];
                if (!mx_big(nq, kn, 2048) || nsplit > nq) nsplit = nq;
                const unsigned nblk_mx = (unsigned)(nsplit * ngrp) + (unsigned)(((int64_t)(nq - nsplit) * ngrp + 3) / 4);
                MxFar mf;
                memset(&mf, 0, sizeof mf);
                mf.nlev = itp.nlev;
                for (int l = 0; l < itp.nlev; l++) { mf.ioff[l] = itp.ioff[l]; mf.nfar[l] = itp.nfar[l] > 0 ? itp.nfar[l] : CS_NC; }
                mf.ioff[itp.nlev] = itp.nItot;
                mf.R = (itp.far_shared_full && nsplit == nq) ? nullptr : itp.R;
                CS_LAUNCH(k_cheb_nodes_mx, dim3(nblk_mx), dim3(256), 0, s, itp.nodes, G.L, hot, itp.sep, itp.nItot, q0, nsplit, kn,
                                   itp.Kpad, ngrp, itp.F, itp.iz, mf);
            }
            if (s != sm) {
                (void)hipEventRecord(fork->ev_join, s);
                fork->pending = true;
                s = sm;
            }
            if (evg) (void)hipEventRecord(evg[2], s);
            ChebApply A0;
            ChebApply &A = defer ? *defer : A0;
            if (!defer) A.ngas = 0;
            A.nlev = itp.nlev;
            for (int l = 0; l < itp.nlev; l++) {
                A.shift[l] = 0;
                for (int r = itp.itv[l] / 64; r > 1; r >>= 1) A.shift[l]++;
                A.ioff[l] = itp.ioff[l];
                A.nc[l] = CS_NC;
                A.noff[l] = itp.ioff[l] * CS_NC;
                A.Cm[l] = itp.Cm[l];
            }
            fuse = defer && itp.fuse_apply && use_edge && defer->ngas == 0;   // (then k_voigt_edge_mx below carries this group's node sums to the grid)
            if (fuse) {
                Afuse = A;
                Afuse.ngas = 1;
                Afuse.l0[0] = itp.l0;
                Afuse.F[0] = itp.F;
            } else if (defer && A.ngas > 0) {
                A.l0[0] = std::min(A.l0[0], itp.l0);   // same F: the sum over the gases so far
            } else {
                A.l0[A.ngas] = itp.l0;
                A.F[A.ngas++] = itp.F;
            }
            if (!defer) {   // sigma = base + extra + interpolated far wings now; the per-point kernels add the rest
                launch_apply_cascade(s, A, itp.Rc, itp.itv, itp.nI, itp.cascade, itp.Kpad, nnu, kn, base, extra, sigma, accumulate);
                accumulate = 1;
            }               // (deferred: the caller applies the node sums of all its gases in one launch, after the last gas)
            const int low = itp.nlev - 1;
            iz = itp.iz + itp.ioff[low];
            ishift = A.shift[low];
        } else if (evg) {
            (void)hipEventRecord(evg[1], s);
            (void)hipEventRecord(evg[2], s);
        }
        const int nblk = (nt64 + 3) / 4;
        g_line_kernel = 0;

        // waves per tile: enough waves to fill 256 CUs x 32 wave slots about 4 times over
        const int64_t nwave = (int64_t)nt64 * kn;
        int split = nwave >= 16384 ? 1 : (nwave >= 4096 ? 2 : 4);   // (re-tuned with the far wings interpolated: waves are 3x shorter)
        if (itp.far_split == 1 || itp.far_split == 2 || itp.far_split == 4) split = itp.far_split;   // (cs_set_tuning key 22, A/B)
        const int nblk_s = (nt64 * split + 3) / 4;
        // 8 x (blocks of the longest XCD stretch): XCD-aware tile mapping (tile_block); xtiles is a multiple of 4 tiles
        const dim3 grid_s((unsigned)(8 * (xtiles * split / 4)), kn);
#define CS_FAR_LAUNCH(MIX, SP) CS_LAUNCH((k_voigt_far<MIX, SP, false>), grid_s, dim3(256), 0, s, dnu, nnu, G.L, hot, hot32, G.nu.as<double>(), \
                                                  win, zones, nt64, nblk_s, cut, base, extra, sigma, accumulate, ranges, iz, itp.nItot, ishift, edgez, zero2)
#define CS_LOR_LAUNCH(SP) CS_LAUNCH((k_voigt_far<false, SP, true>), grid_s, dim3(256), 0, s, dnu, nnu, G.L, hot, hot32, G.nu.as<double>(), \
                                                  win, zones, nt64, nblk_s, cut, base, extra, sigma, accumulate, ranges, iz, itp.nItot, ishift, edgez)
        const EdgeZone *edgez = use_edge ? itp.edge : nullptr;
        // near-line kernels and the sub-tile cores on a side stream, into their own plane (Voigt only; not while profiling)
        const bool near_fork = !lor && fork && fork->use_near && fork->sigma2 && defer && !evg;
        double *zero2 = nullptr;   // (the far kernel can clear the plane itself: unused since k_voigt_sub adds to it beside k_voigt_far)
        if (near_fork) {   // zones, records and piece tables are written: the side stream may start
            if (forked_here) {   // (nothing was enqueued on the main stream since that record: one event serves both side streams)
                (void)hipStreamWaitEvent(fork->s3, fork->ev_fork, 0);
            } else {
                (void)hipEventRecord(fork->ev_fork3, s);
                (void)hipStreamWaitEvent(fork->s3, fork->ev_fork3, 0);
            }
            // the plane's first writer of the step defines all of it: k_voigt_sub where it runs (sums where a tile has a core, zeros
            // elsewhere: cs_set_tuning key 19 = 1 keeps the memset for A/B), else a memset
            const bool sub_here = use_edge && itp.core;
            const bool sub_assigns = sub_here && !fork->zeroed && !itp.near_memset;
            if (!fork->zeroed && !sub_assigns) (void)hipMemsetAsync(fork->sigma2, 0, (size_t)kn * nnu * sizeof(double), fork->s3);
            fork->zeroed = true;
            if (sub_here)
                CS_LAUNCH(k_voigt_sub<CS_SUBW>, dim3((unsigned)nt64, (unsigned)((kn + 64 / CS_SUBW - 1) / (64 / CS_SUBW))), dim3(4096 / CS_SUBW), 0, fork->s3, dnu, nnu, G.L, hot,
                          G.nu.as<double>(), zones, itp.edge, nt64, kn, cut, fork->sigma2, reinterpret_cast<unsigned *>(ranges), near_prio, sub_assigns ? 1 : 0);
        }
        if (lor) {
            if (split == 1) CS_LOR_LAUNCH(1); else if (split == 2) CS_LOR_LAUNCH(2); else CS_LOR_LAUNCH(4);
        } else if (hot32 && use_edge) {
#define CS_EDGE32_LAUNCH(SP) CS_LAUNCH((k_voigt_far<true, SP, false, true>), grid_s, dim3(256), 0, s, dnu, nnu, G.L, hot, hot32, G.nu.as<double>(), \
                                                  win, zones, nt64, nblk_s, cut, base, extra, sigma, accumulate, ranges, iz, itp.nItot, ishift, edgez, zero2)
            if (split == 1) CS_EDGE32_LAUNCH(1); else if (split == 2) CS_EDGE32_LAUNCH(2); else CS_EDGE32_LAUNCH(4);
#undef CS_EDGE32_LAUNCH
        } else if (hot32) {
            if (split == 1) CS_FAR_LAUNCH(true, 1); else if (split == 2) CS_FAR_LAUNCH(true, 2); else CS_FAR_LAUNCH(true, 4);
        } else if (use_edge) {
#define CS_EDGE_LAUNCH(SP) CS_LAUNCH((k_voigt_far<false, SP, false, true>), grid_s, dim3(256), 0, s, dnu, nnu, G.L, hot, hot32, G.nu.as<double>(), \
                                                  win, zones, nt64, nblk_s, cut, base, extra, sigma, accumulate, ranges, iz, itp.nItot, ishift, edgez, zero2)
            if (split == 1) CS_EDGE_LAUNCH(1); else if (split == 2) CS_EDGE_LAUNCH(2); else CS_EDGE_LAUNCH(4);
#undef CS_EDGE_LAUNCH
        } else {
            if (split == 1) CS_FAR_LAUNCH(false, 1); else if (split == 2) CS_FAR_LAUNCH(false, 2); else CS_FAR_LAUNCH(false, 4);
        }
#undef CS_FAR_LAUNCH
#undef CS_LOR_LAUNCH
        if (evg) (void)hipEventRecord(evg[3], s);
        if (use_edge && itp.core && !near_fork)   // the window cores of the groups whose series radius is short: pairs inside it (the rest: k_voigt_edge_mx)
            CS_LAUNCH(k_voigt_sub<CS_SUBW>, dim3((unsigned)nt64, (unsigned)((kn + 64 / CS_SUBW - 1) / (64 / CS_SUBW))), dim3(4096 / CS_SUBW), 0, s, dnu, nnu, G.L, hot, G.nu.as<double>(), zones,
                               itp.edge, nt64, kn, cut, sigma, reinterpret_cast<unsigned *>(ranges), near_prio, 0);
        auto launch_near = [&](hipStream_t sn, double *out) {
            const int ngrpn = (nt64 + CS_NEAR_R - 1) / CS_NEAR_R;   // near kernels: one wave = CS_NEAR_R consecutive tiles ...
            // ... times nrep, one after the other: where the table is sparse against the grid (few tiles have candidates at all) and the
            // grid long enough to keep the chip full with an eighth of the waves, and on every grid of half a million (tile, state) waves
            // and more -- there the launch of the waves is what a near-line kernel costs first (BASELINE configs[4]: 7.9e5 waves per
            // tier, step 7.06 -> 6.85 ms with eight tiles per wave; the bench column, 9.5e4 waves: a tie with two, a loss with four)
            const int64_t nwaves_near = (int64_t)ngrpn * kn;
            const int nrep = (nwaves_near >= 524288 || (jhi - jlo < (int64_t)nt64 * 2 && nwaves_near >= 262144)) ? 8 : 1;
            const dim3 gridq((unsigned)(((ngrpn + nrep - 1) / nrep + 3) / 4), kn);
            if (nrep == 1 && itp.near_both) {   // both tiers in one launch (one tile per wave; cs_set_tuning key 16 | 4: two launches, A/B)
                CS_LAUNCH(k_voigt_near_both, gridq, dim3(256), 0, sn, dnu, nnu, G.L, hot, cold, zones, nt64, cut, out, ranges, near_prio);
                g_near_launches += 1;
                return;
            }
            g_near_launches += 2;
            CS_LAUNCH(k_voigt_near<0>, gridq, dim3(256), 0, sn, dnu, nnu, G.L, hot, cold, zones, nt64, ngrpn, nrep, cut, out, ranges, near_prio);
            CS_LAUNCH(k_voigt_near<1>, gridq, dim3(256), 0, sn, dnu, nnu, G.L, hot, cold, zones, nt64, ngrpn, nrep, cut, out, ranges, near_prio);
        };
        if (near_fork) {   // the near kernels need the hand-off words of both k_voigt_far (main stream) and k_voigt_sub (theirs)
            (void)hipEventRecord(fork->ev_far3, s);
            (void)hipStreamWaitEvent(fork->s3, fork->ev_far3, 0);
            launch_near(fork->s3, fork->sigma2);
            (void)hipEventRecord(fork->ev_join3, fork->s3);
            fork->pending3 = true;
            fork->live = true;
        }
        if (evg) (void)hipEventRecord(evg[4], s);
        if (use_edge)
        {
            if (fuse) fork_join(fork, s);   // (it reads F)
            if (mx_big(nt64, kn, 1024))
                CS_LAUNCH(k_voigt_edge_mx<1>, dim3((unsigned)((nt64 + 3) / 4), (unsigned)((kn + 15) / 16)), dim3(256), 0, s, dnu, nnu, G.L, hot, win,
                          itp.edge, nt64, kn, cut, sigma, fuse ? 1 : 0, Afuse, itp.Kpad, G.nu.as<double>(), itp.edge_phases,
                          itp.edge_phases ? itp.tnodes : nullptr, itp.tC);
            else   // short grid: four waves per (tile, group)
                CS_LAUNCH(k_voigt_edge_mx<4>, dim3((unsigned)nt64, (unsigned)((kn + 15) / 16)), dim3(256), 0, s, dnu, nnu, G.L, hot, win,
                          itp.edge, nt64, kn, cut, sigma, fuse ? 1 : 0, Afuse, itp.Kpad, G.nu.as<double>(), itp.edge_phases,
                          (const double *)nullptr, (const double *)nullptr);   // (the 16-node path in the shared form: 16 more matrix steps per WAVE, no gain measured)
        }
        if (evg) (void)hipEventRecord(evg[5], s);
        if (!lor && !near_fork) launch_near(s, sigma);
    } else if (shape == SH_PHCO2 && ph && phco2_fast_ok(G, nnu, cut, kn, ph)) {
        // PHCO2 fast path (k_phco2): region-uniform far lines with factorised chi; needs the cut-off edges inside region 3 and the
        // near zone inside the chi = 1 core (phco2_fast_ok), else the generic kernel below
        const int nt64 = (int)((nnu + 63) / 64);
        pa.phfac = ph->fac.as<double>();
        pa.nu_c = ph->nu_c;
        ZoneArgs za;
        za.nu = dnu; za.nul = G.nu.as<double>(); za.Tk = Tk; za.gbound = gbound; za.win = win; za.zones = zones; za.nnu = nnu;
        za.lorentz = 0;
        za.ntile = nt64; za.K = kn; za.mu_min = G.mu_min; za.mu_max = G.mu_max; za.cut = cut; za.far_s = far_s;
        za.margin = kChebMargin;
        const unsigned nb_zones = (unsigned)(((int64_t)nt64 * kn + 255) / 256);
        IzParams P;
        memset(&P, 0, sizeof P);
        CS_LAUNCH(k_gas_setup, dim3(nb_prep + nb_zones), dim3(256), 0, s, nb_prep, nb_zones, pa, za, P, (IZone *)nullptr);
        const bool use_itp = ph_interp_ready(ph, dnu, nnu, cut, kn, s);
        const PhScratch::Grid &pg = ph->grid;
        PhArgs pw;
        pw.nu = dnu; pw.nul = G.nu.as<double>(); pw.nnu = nnu; pw.ntile = nt64; pw.J0 = (int32_t)jlo; pw.J1 = (int32_t)jhi; pw.cut = cut;
        pw.tol = 1e-9 * (std::max(std::fabs(ph->nu_lo), std::fabs(ph->nu_hi)) + cut + 1.0);
        pw.out = ph->win.as<PhWin>();
        PhIArgs ia;
        memset(&ia, 0, sizeof ia);
        if (use_itp) {
            ia.nu = dnu; ia.nul = pw.nul; ia.nnu = nnu; ia.J0 = pw.J0; ia.J1 = pw.J1; ia.cut = cut; ia.tol = pw.tol; ia.margin = ph->margin;
            ia.lv = pg.lv;
            ia.out = ph->piw.as<PhIWin>();
        }
        // the pairs within 3 cm^-1 (chi = 1: plain Voigt, every near-line pair among them) through the Voigt kernels, where their
        // near-line hand-off takes this table (check_near_density; else k_phco2's own core loop)
        bool inner = false;
        if (ranges && !ph->own_core) {
            const PhScratch::Dens *hit = nullptr;
            for (auto &d : ph->dens) if (d.tab == (const void *)&G && d.gen == G.generation && d.grid == ph->grid_id) hit = &d;
            if (!hit) {
                if (ph->dens.size() >= 8) ph->dens.erase(ph->dens.begin());
                ph->dens.push_back({(const void *)&G, G.generation, ph->grid_id, check_near_density(G, ph->nu_hi, ph->max_span, 3.0) == CS_OK});
                hit = &ph->dens.back();
            }
            inner = hit->ok && ph->win3.reserve((size_t)(nt64 + 3) * sizeof(WaveWin)) == hipSuccess &&
                    ph->zones3.reserve((size_t)kn * nt64 * sizeof(Zone)) == hipSuccess;
        }
        WwArgs wa;
        memset(&wa, 0, sizeof wa);
        if (inner) {
            wa.nu = dnu; wa.nul = pw.nul; wa.nnu = nnu; wa.ntile = nt64; wa.J0 = pw.J0; wa.J1 = pw.J1; wa.cut = 3.0;
            wa.sparse = (jhi - jlo) * 8 < nnu ? 1 : 0;
            wa.out = ph->win3.as<WaveWin>();
        }
        const unsigned nb_tiles = (unsigned)((nt64 + 255) / 256), nb_itv = (unsigned)((ia.lv.nItot + 255) / 256);
        CS_LAUNCH(k_phwin, dim3(nb_tiles + nb_itv + (inner ? nb_tiles : 0u)), dim3(256), 0, s, nb_tiles, nb_itv, pw, ia, wa);
        if (evg) (void)hipEventRecord(evg[0], s);
        const PhIWin *piw = nullptr;
        PhFine fine;
        memset(&fine, 0, sizeof fine);
        if (use_itp) {   // far wings of the region-uniform lines: node sums, carried to the grid (sigma = base + extra + them)
            const int Kpad = cheb_kpad(kn);
            CS_LAUNCH(k_phco2_nodes, dim3((unsigned)pg.nslots * (unsigned)((kn + 3) / 4)), dim3(256), 0, s, pg.nodes.as<double>(), G.L, hot,
                      ph->fac.as<double>(), ph->nu_c, ph->piw.as<PhIWin>(), pg.lv, pg.vl, kn, Kpad, Tk, cut, gbound, G.mu_min, G.mu_max, far_s,
                      ph->F.as<double>());
            if (evg) (void)hipEventRecord(evg[1], s);
            ChebApply A;
            memset(&A, 0, sizeof A);
            A.nlev = pg.vl.nv; A.ngas = 1; A.F[0] = ph->F.as<double>(); A.l0[0] = 0;
            for (int v = 0; v < pg.vl.nv; v++) {
                for (int r = pg.lv.itv[pg.vl.rl[v]] / 64; r > 1; r >>= 1) A.shift[v]++;
                A.ioff[v] = pg.vl.boff[v];
                A.nc[v] = pg.vl.nc[v];
                A.noff[v] = pg.vl.noff[v];
                A.Cm[v] = pg.Cm[v].as<double>();
            }
            launch_apply(s, A, Kpad, nnu, kn, base, extra, sigma, accumulate, true);
            accumulate = 1;
            if (evg) (void)hipEventRecord(evg[2], s);
            piw = ph->piw.as<PhIWin>();
            fine = pg.fine;
        } else if (evg) { (void)hipEventRecord(evg[1], s); (void)hipEventRecord(evg[2], s); }
        g_line_kernel = 2;
        CS_LAUNCH(k_phco2, dim3((unsigned)((nt64 + 3) / 4), kn), dim3(256), 0, s, dnu, nnu, G.L, hot, cold, ph->fac.as<double>(), ph->nu_c,
                           ph->win.as<PhWin>(), zones, nt64, cut, Tk, kn, base, extra, sigma, accumulate, piw, fine, inner ? 1 : 0);
        if (evg) (void)hipEventRecord(evg[3], s);
        if (inner) {
            const int nt4 = (nt64 + 3) / 4 * 4, per = ((nt4 / 4 + 7) / 8) * 4;   // (wave_windows' stretch length)
            launch_gas(s, SH_VOIGT, G, jrange0, jrange1, kn, Tk, Pk, Ppk, scale, mstride, lrt, qrefq, hot, cold, dnu, nnu, ntile256, J0, J1,
                       ph->win3.as<WaveWin>(), per, ph->zones3.as<Zone>(), ranges, gbound, 3.0, 0.0, nullptr, sigma, 1, nullptr, nullptr, far_s,
                       Interp(), nullptr, nullptr, nullptr, true);
            g_line_kernel = 2;   // (the group's far lines were k_phco2's; the inner pass only took the pairs within 3 cm^-1)
        }
        if (evg) { (void)hipEventRecord(evg[4], s); (void)hipEventRecord(evg[5], s); }
    } else {
        if (nb_prep > 0) {
            ZoneArgs za;
            IzParams P;
            memset(&za, 0, sizeof za);
            memset(&P, 0, sizeof P);
            CS_LAUNCH(k_gas_setup, dim3(nb_prep), dim3(256), 0, s, nb_prep, 0u, pa, za, P, (IZone *)nullptr);
        }
        if (evg) { (void)hipEventRecord(evg[0], s); (void)hipEventRecord(evg[1], s); (void)hipEventRecord(evg[2], s); }
        g_line_kernel = 1;
        launch_linesum_shape(shape, dim3(ntile256, kn), s, dnu, nnu, G.L, hot, cold, J0, J1, cut, Tk, base, extra, sigma,
                             accumulate);
        if (evg) { (void)hipEventRecord(evg[3], s); (void)hipEventRecord(evg[4], s); (void)hipEventRecord(evg[5], s); }
    }
}

int check_gas_states(const GasTable &G, int K, const double *T)
{
    for (int k = 0; k < K; k++)
        if (!(T[k] >= kTmin && T[k] <= kTmax))
            return fail(CS_ETEMP, "temperature %g K outside of Qref/Q interpolation range [25, 1000]", T[k]);
    for (int64_t j = 0; j < G.L; j++) {
        int I = G.h_iso[j];
        if (I < 1 || I > G.niso || G.h_ncheb[I - 1] <= 0)
            return fail(CS_ENOCHEB, "no interpolating polynomial available to compute Qref/Q for isotopologue %d", I);
    }
    return CS_OK;
}

int check_ascending(const double *nu, int64_t n)
{
    if (n < 1) return fail(CS_EINVAL, "empty wavenumber vector");
    for (int64_t i = 1; i < n; i++)
        if (!(nu[i] > nu[i - 1])) return fail(CS_EORDER, "wavenumber vectors must be sorted in ascending order");
    return CS_OK;
}

}  // namespace

extern "C" {

int cs_version(void) { return 100; }
#ifndef CS_BUILD_ID
#define CS_BUILD_ID "unknown"
#endif
const char *cs_build_id(void) { return CS_BUILD_ID; }
const char *cs_last_error(void) { return g_err.c_str(); }

int cs_create(int device, cs_ctx **out)
{
    if (!out) return fail(CS_EINVAL, "out is NULL");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(CS_EINVAL, "device %d out of range (%d devices)", device, n);
    HIPCHK(hipSetDevice(device));
    cs_ctx *c = new cs_ctx();
    c->device = device;
    {   // the code object is built for gfx950 only; k_flux_*'s in-kernel band sum (flux_last_block_reduce) additionally relies on a
        // property of that architecture -- see its comment -- so it is switched on by the device's name, not assumed
        hipDeviceProp_t prop;
        c->gfx950 = hipGetDeviceProperties(&prop, device) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork3, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join3, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_far3, hipEventDisableTiming);
    if (e != hipSuccess) { cs_destroy(c); return fail(CS_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    {   // R_n[m][j] = l_j(x_m): Lagrange basis of the n extrema cos(pi j / (n - 1)) at the 64 extrema cos(pi m / 63), barycentric form
        std::vector<double> R((size_t)CS_NC * 48);
        const long double pi = 3.14159265358979323846264338327950288L;
        for (int n : {32, 16}) {
            double *Rn = R.data() + (n == 32 ? 0 : CS_NC * 32);
            std::vector<long double> x(n), w(n);
            for (int j = 0; j < n; j++) { x[j] = cosl(pi * j / (n - 1)); w[j] = ((j & 1) ? -1.0L : 1.0L) * ((j == 0 || j == n - 1) ? 0.5L : 1.0L); }
            for (int m = 0; m < CS_NC; m++) {
                const long double xm = cosl(pi * m / (CS_NC - 1));
                int hit = -1;
                long double den = 0.0L;
                for (int j = 0; j < n; j++) {
                    const long double d = xm - x[j];
                    if (fabsl(d) < 1e-17L) hit = j; else den += w[j] / d;
                }
                for (int j = 0; j < n; j++)
                    Rn[(size_t)m * n + j] = hit >= 0 ? (j == hit ? 1.0 : 0.0) : (double)((w[j] / (xm - x[j])) / den);
            }
        }
        if (upload(c->reinterp, R.data(), R.size(), c->stream) != CS_OK || hipStreamSynchronize(c->stream) != hipSuccess) {
            cs_destroy(c);
            return fail(CS_EHIP, "upload of the re-interpolation matrices failed");
        }
    }
    *out = c;
    c->counted = true;
    g_live_ctx++;
    return CS_OK;
}

void cs_destroy(cs_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->counted && --g_live_ctx == 0) pool_stop();
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    drop_graph(ctx->col);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_fork3) (void)hipEventDestroy(ctx->ev_fork3);
    if (ctx->ev_join3) (void)hipEventDestroy(ctx->ev_join3);
    if (ctx->ev_far3) (void)hipEventDestroy(ctx->ev_far3);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// host mirror -> statistics + device arrays
static int table_to_device(GasTable &G, hipStream_t s)
{
    const int64_t L = G.L;
    G.mu_min = *std::min_element(G.h_mu.begin(), G.h_mu.end());
    G.mu_max = *std::max_element(G.h_mu.begin(), G.h_mu.end());
    G.ga_max = *std::max_element(G.h_ga.begin(), G.h_ga.end());
    G.gs_max = *std::max_element(G.h_gs.begin(), G.h_gs.end());
    G.na_min = *std::min_element(G.h_na.begin(), G.h_na.end());
    G.na_max = *std::max_element(G.h_na.begin(), G.h_na.end());
    if (!(G.mu_min > 0)) return fail(CS_EINVAL, "isotopologue molar masses must be positive");
    std::vector<double> sref(L);   // scaleintensity, line_shapes.jl:107-123: the denominator at Tref does not depend on the state
    for (int64_t j = 0; j < L; j++) sref[j] = G.h_S[j] / (std::exp(-kC2 * G.h_Epp[j] / kTref) * (1.0 - std::exp(-kC2 * G.h_nu[j] / kTref)));
    int rc;
    if ((rc = upload(G.sref, sref.data(), L, s))) return rc;
    if ((rc = upload(G.nu, G.h_nu.data(), L, s)) || (rc = upload(G.S, G.h_S.data(), L, s)) || (rc = upload(G.ga, G.h_ga.data(), L, s)) ||
        (rc = upload(G.gs, G.h_gs.data(), L, s)) || (rc = upload(G.Epp, G.h_Epp.data(), L, s)) || (rc = upload(G.na, G.h_na.data(), L, s)) ||
        (rc = upload(G.mu, G.h_mu.data(), L, s)) || (rc = upload(G.iso, G.h_iso.data(), L, s)) || (rc = upload(G.ncheb, G.h_ncheb.data(), G.niso, s)) ||
        (rc = upload(G.cheb, G.h_cheb.data(), (size_t)G.niso * CS_CHEB_LD, s)))
        return rc;
    if (!G.h_gid.empty() && (rc = upload(G.gid, G.h_gid.data(), L, s))) return rc;
    HIPCHK(hipStreamSynchronize(s));   // (sref is a local)
    G.present = true;
    return CS_OK;
}

static uint64_t next_generation()   // (cs_fluxes_discretized_multi sets columns up on one host thread per context)
{
    static std::atomic<uint64_t> g{0};
    return ++g;
}

int cs_gas_upload(cs_ctx *ctx, int slot, int64_t L, const double *nu, const double *S, const double *gamma_a,
                  const double *gamma_s, const double *Epp, const double *na, const double *mu_iso,
                  const int16_t *iso, int niso, const int32_t *ncheb, const double *cheb)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    if (slot < 0 || slot >= CS_MAX_GAS) return fail(CS_EINVAL, "gas slot %d out of range", slot);
    if (L < 1 || niso < 1) return fail(CS_EINVAL, "empty line table");
    for (int64_t j = 1; j < L; j++)
        if (!(nu[j] >= nu[j - 1])) return fail(CS_EORDER, "line table must be sorted by wavenumber (par.jl:267)");
    // line indices travel as int32 (windows, zones) and as 26-bit fields of the near-line queue entries (k_voigt_near)
    if (L >= ((int64_t)1 << 26)) return fail(CS_EINVAL, "line table too long (%lld lines; the limit is 2^26 - 1 per gas)", (long long)L);
    HIPCHK(hipSetDevice(ctx->device));
    GasTable &G = ctx->gas[slot];
    G.present = false;
    G.L = L;
    G.niso = niso;
    G.h_nu.assign(nu, nu + L); G.h_S.assign(S, S + L); G.h_ga.assign(gamma_a, gamma_a + L); G.h_gs.assign(gamma_s, gamma_s + L);
    G.h_Epp.assign(Epp, Epp + L); G.h_na.assign(na, na + L); G.h_mu.assign(mu_iso, mu_iso + L);
    G.h_iso.assign(iso, iso + L);
    G.h_ncheb.assign(ncheb, ncheb + niso);
    G.h_cheb.assign(cheb, cheb + (size_t)niso * CS_CHEB_LD);
    G.h_gid.clear();
    G.members.clear();
    int rc;
    if ((rc = table_to_device(G, ctx->stream))) return rc;
    G.generation = next_generation();
    return CS_OK;
}

// One sorted table out of the tables of `slots` (stable merge by wavenumber: lines of equal position keep member order), each
// line tagged with its member index; isotopologue numbers are offset into the concatenated Chebyshev tables.  Kept by the context
// (a handful, keyed by the members' upload generations) so that re-setting a column up does not merge again.
static int merged_table(cs_ctx *ctx, const std::vector<int> &slots, std::shared_ptr<const GasTable> *out)
{
    std::vector<std::pair<int, uint64_t>> key;
    for (int sl : slots) key.emplace_back(sl, ctx->gas[sl].generation);
    for (size_t i = 0; i < ctx->merged.size(); i++)
        if (ctx->merged[i]->members == key) {   // a hit becomes the most recently used entry
            std::shared_ptr<GasTable> m = ctx->merged[i];
            ctx->merged.erase(ctx->merged.begin() + i);
            ctx->merged.push_back(m);
            *out = m;
            return CS_OK;
        }
    // the cache only bounds what the CONTEXT keeps: a column shares ownership of the tables of its launch groups (ColGas::hold), so
    // evicting the least recently used entry can never free a table a column still runs on, however many groups that column has
    if (ctx->merged.size() >= CS_MAX_GAS / 2) ctx->merged.erase(ctx->merged.begin());
    int64_t L = 0;
    int niso = 0;
    for (int sl : slots) { L += ctx->gas[sl].L; niso += ctx->gas[sl].niso; }
    if (L >= ((int64_t)1 << 26)) return fail(CS_EINVAL, "merged line table too long (%lld lines; the limit is 2^26 - 1)", (long long)L);
    std::shared_ptr<GasTable> M(new GasTable());
    GasTable &G = *M;
    G.L = L;
    G.niso = niso;
    G.members = key;
    struct Src { uint8_t m; int32_t j; };
    std::vector<Src> order, next, mine;
    std::vector<int> iso_off(slots.size());
    int off = 0;
    auto nu_of = [&](const Src &q) { return ctx->gas[slots[q.m]].h_nu[q.j]; };
    for (size_t m = 0; m < slots.size(); m++) {   // members are sorted: merge them one by one (stable: equal positions keep member order)
        const GasTable &g = ctx->gas[slots[m]];
        iso_off[m] = off;
        off += g.niso;
        mine.resize((size_t)g.L);
        for (int64_t j = 0; j < g.L; j++) mine[j] = Src{(uint8_t)m, (int32_t)j};
        next.resize(order.size() + mine.size());
        std::merge(order.begin(), order.end(), mine.begin(), mine.end(), next.begin(), [&](const Src &a, const Src &b) { return nu_of(a) < nu_of(b); });
        order.swap(next);
        G.h_ncheb.insert(G.h_ncheb.end(), g.h_ncheb.begin(), g.h_ncheb.end());
        G.h_cheb.insert(G.h_cheb.end(), g.h_cheb.begin(), g.h_cheb.end());
    }
    G.h_nu.resize(L); G.h_S.resize(L); G.h_ga.resize(L); G.h_gs.resize(L); G.h_Epp.resize(L); G.h_na.resize(L); G.h_mu.resize(L);
    G.h_iso.resize(L); G.h_gid.resize(L);
    for (int64_t i = 0; i < L; i++) {
        const Src q = order[i];
        const GasTable &g = ctx->gas[slots[q.m]];
        G.h_nu[i] = g.h_nu[q.j]; G.h_S[i] = g.h_S[q.j]; G.h_ga[i] = g.h_ga[q.j]; G.h_gs[i] = g.h_gs[q.j];
        G.h_Epp[i] = g.h_Epp[q.j]; G.h_na[i] = g.h_na[q.j]; G.h_mu[i] = g.h_mu[q.j];
        G.h_iso[i] = (int16_t)(g.h_iso[q.j] + iso_off[q.m]);
        G.h_gid[i] = q.m;
    }
    int rc;
    if ((rc = table_to_device(G, ctx->stream))) return rc;
    G.generation = next_generation();
    *out = M;
    ctx->merged.push_back(std::move(M));
    return CS_OK;
}

int cs_set_precision(cs_ctx *ctx, int mode, double far_s)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    if (mode != 0 && mode != 1) return fail(CS_EINVAL, "precision mode must be 0 (fp64) or 1 (fp32 far wings)");
    if (!(far_s >= 1e6)) return fail(CS_EINVAL, "far_s must be >= 1e6");
    ctx->mixed = mode;
    ctx->far_s = far_s;
    drop_graph(ctx->col);
    return CS_OK;
}

int cs_set_interp(cs_ctx *ctx, int on)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    ctx->interp = on ? 1 : 0;
    return CS_OK;
}

int cs_set_interp_plan(cs_ctx *ctx, int first_level, int size_min, int size_max)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    if (size_min < 128 || size_max > 2048 || size_min > size_max) return fail(CS_EINVAL, "interval sizes must satisfy 128 <= size_min <= size_max <= 2048");
    if (first_level < -1 || first_level > CS_MAX_LEVEL) return fail(CS_EINVAL, "first_level must be -1 (automatic) or 0..%d", CS_MAX_LEVEL);
    ctx->itp_first = first_level;
    ctx->itp_min = size_min;
    ctx->itp_max = size_max;
    return CS_OK;
}

int cs_set_matrix_cores(cs_ctx *ctx, int on)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    if (on < 0) on = 0;
    ctx->matrix_core = (on & 4) ? 0 : 1;   // (tuning / tests: | 4 keeps the tile-wide near-zone pass everywhere)
    on &= 3;
    ctx->matrix_nodes = on > 2 ? 2 : on;
    drop_graph(ctx->col);
    return CS_OK;
}

int cs_set_merge(cs_ctx *ctx, int on)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    ctx->merge = on ? 1 : 0;
    return CS_OK;
}

static int interp_key(const cs_ctx *ctx)
{
    return (ctx->tune[5] ? 65536 : 0) + ctx->interp * 4096 + (ctx->itp_first + 1) * 256 + (ctx->itp_min >> 7) * 16 + (ctx->itp_max >> 7);
}

int cs_set_tuning(cs_ctx *ctx, int key, int value)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    if (key < 0 || key >= CS_NTUNE) return fail(CS_EINVAL, "tuning key %d out of range", key);
    if (key == 3 && value != 0 && (value < 15 || value > 100)) return fail(CS_EINVAL, "interpolation margin must be 15..100 per cent of the half-width");
    ctx->tune[key] = value;
    drop_graph(ctx->col);
    return CS_OK;
}

int cs_gas_clear(cs_ctx *ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= CS_MAX_GAS) return fail(CS_EINVAL, "bad slot");
    for (auto &g : ctx->col.ugas)
        if (g.slot == slot) ctx->col.ready = false;   // (the resident column's windows belong to the table that goes away)
    ctx->gas[slot] = GasTable();
    return CS_OK;
}

static int shape_impl(cs_ctx *ctx, int slot, int shape, double dnu_cut, int64_t nnu, const double *nu, int K,
                      const double *T, const double *P, const double *Pp, double *sigma, int64_t ld_state, bool strict);

int cs_shape_batch(cs_ctx *ctx, int slot, int shape, double dnu_cut, int64_t nnu, const double *nu, int K,
                   const double *T, const double *P, const double *Pp, double *sigma, int64_t ld_state)
{
    return shape_impl(ctx, slot, shape, dnu_cut, nnu, nu, K, T, P, Pp, sigma, ld_state, true);
}

int cs_shape_points(cs_ctx *ctx, int slot, int shape, double dnu_cut, int64_t nnu, const double *nu, int K,
                    const double *T, const double *P, const double *Pp, double *sigma, int64_t ld_state)
{
    return shape_impl(ctx, slot, shape, dnu_cut, nnu, nu, K, T, P, Pp, sigma, ld_state, false);
}

static int shape_impl(cs_ctx *ctx, int slot, int shape, double dnu_cut, int64_t nnu, const double *nu, int K,
                      const double *T, const double *P, const double *Pp, double *sigma, int64_t ld_state, bool strict)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    if (slot < 0 || slot >= CS_MAX_GAS || !ctx->gas[slot].present) return fail(CS_EINVAL, "gas slot %d is empty", slot);
    if (shape < 0 || shape > 3) return fail(CS_EINVAL, "unknown shape %d", shape);
    if (K < 1 || ld_state < nnu) return fail(CS_EINVAL, "bad K/ld_state");
    int rc;
    if ((rc = check_ascending(nu, nnu))) return rc;
    GasTable &G = ctx->gas[slot];
    if ((rc = check_gas_states(G, K, T))) return rc;
    if (shape == SH_VOIGT && (rc = check_near_density(G, nu, nnu, dnu_cut))) return rc;
    ph_set_grid(ctx, ctx->ph, nu, nnu, ++g_grid_counter);
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int64_t g0, g1, pairs, inr;
    included_range(G.h_nu, nu[0], nu[nnu - 1], dnu_cut, strict, g0, g1);
    std::vector<int32_t> J0, J1;
    tile_windows(G.h_nu, g0, g1, nu, nnu, window_reach(shape, G, nu[nnu - 1], dnu_cut), J0, J1, pairs, inr);
    const int ntile = (int)J0.size();
    DevBuf dnu, dT, dP, dPp, dJ0, dJ1, hot, cold, dsig, dwin, dzones, dgmax, dranges, dlrt, dqref;
    {
        std::vector<double> lrt, qr;
        state_tables(G, K, T, lrt, qr);
        if ((rc = upload(dlrt, lrt.data(), lrt.size(), s)) || (rc = upload(dqref, qr.data(), qr.size(), s))) return rc;
        HIPCHK(hipStreamSynchronize(s));
    }
    std::vector<WaveWin> win;
    const int xtiles = wave_windows(G.h_nu, g0, g1, nu, nnu, dnu_cut, win);
    if ((rc = upload(dnu, nu, nnu, s)) || (rc = upload(dT, T, K, s)) || (rc = upload(dP, P, K, s)) ||
        (rc = upload(dPp, Pp, K, s)) || (rc = upload(dJ0, J0.data(), ntile, s)) || (rc = upload(dJ1, J1.data(), ntile, s)) ||
        (rc = upload(dwin, win.data(), win.size(), s)))
        return rc;
    // bound the workspace: process the states in chunks
    const size_t per_state = (size_t)G.L * (sizeof(LineHot) + sizeof(LineCold)) + (size_t)nnu * (sizeof(double) + sizeof(int2));
    int kc = (int)std::max<size_t>(1, std::min<size_t>({(size_t)K, ((size_t)4 << 30) / per_state, (size_t)65535}));   // gridDim.y limit
    HIPCHK(hot.reserve(((size_t)kc * G.L + 4) * sizeof(LineHot)));
    HIPCHK(cold.reserve((size_t)kc * G.L * sizeof(LineCold)));
    HIPCHK(dsig.reserve((size_t)kc * nnu * sizeof(double)));
    HIPCHK(dzones.reserve((size_t)kc * win.size() * sizeof(Zone)));
    HIPCHK(dgmax.reserve((size_t)K * sizeof(double)));
    LineF32 *mix32 = nullptr;
    if (ctx->mixed && shape == SH_VOIGT) {
        HIPCHK(ctx->hot32.reserve(((size_t)kc * G.L + 4) * sizeof(LineF32)));
        mix32 = ctx->hot32.as<LineF32>();
    }
    HIPCHK(dranges.reserve((size_t)kc * nnu * sizeof(int2) + (size_t)2 * kc * ((nnu + 63) / 64) * sizeof(int)));   // + per-(tile, state) flags
    {
        std::vector<double> gb = gamma_bound(G, K, T, P, Pp);
        if ((rc = upload(dgmax, gb.data(), K, s))) return rc;
    }
    ChebGrid cheb;
    GasInterp ginterp;
    Interp itp;
    if (ctx->interp && (shape == SH_VOIGT || shape == SH_LORENTZ)) {
        if ((rc = cheb_build(ctx, cheb, nu, dnu.as<double>(), nnu, dnu_cut, s)) ||
            (rc = gas_interp_build(ctx, ginterp, cheb, G.h_nu, g0, g1, nu, nnu, dnu_cut, kc, s)))
            return rc;
        itp = interp_view(cheb, ginterp, kc);
        if (!ctx->matrix_nodes) itp.sep = nullptr, itp.edge = nullptr;
        itp.sep_always = ctx->matrix_nodes == 2;
        interp_settings(ctx, itp);
        itp.core = ctx->matrix_core != 0;
    }
    for (int k0 = 0; k0 < K; k0 += kc) {
        const int kn = std::min(kc, K - k0);
        launch_gas(s, shape, G, J0.front(), J1.back(), kn, dT.as<double>() + k0, dP.as<double>() + k0, dPp.as<double>() + k0, nullptr, 0,
                   dlrt.as<double>() + k0, dqref.as<double>() + (size_t)k0 * G.niso, hot.as<LineHot>(), cold.as<LineCold>(), dnu.as<double>(), nnu, ntile, dJ0.as<int32_t>(), dJ1.as<int32_t>(),
                   dwin.as<WaveWin>(), xtiles, dzones.as<Zone>(), dranges.as<int2>(), dgmax.as<double>() + k0, dnu_cut, 0.0, nullptr, dsig.as<double>(), 0, nullptr,
                   mix32, ctx->far_s, itp, nullptr, &ctx->ph);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpy2DAsync(sigma + (size_t)k0 * ld_state, ld_state * sizeof(double), dsig.p, nnu * sizeof(double),
                                nnu * sizeof(double), kn, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    return CS_OK;
}

int cs_bake(cs_ctx *ctx, int gas_slot, int table_slot, int shape, double dnu_cut, int64_t nnu, const double *nu, int nT,
            const double *T, int nP, const double *P, const double *conc, double *lnsigma_out)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    if (gas_slot < 0 || gas_slot >= CS_MAX_GAS || !ctx->gas[gas_slot].present) return fail(CS_EINVAL, "gas slot %d is empty", gas_slot);
    if (table_slot < 0 || table_slot >= CS_MAX_TABLE) return fail(CS_EINVAL, "table slot %d out of range", table_slot);
    if (shape < 0 || shape > 3) return fail(CS_EINVAL, "unknown shape %d", shape);
    if (nT < 2 || nP < 2) return fail(CS_EINVAL, "need at least 2 x 2 grid points");
    int rc;
    if ((rc = check_ascending(nu, nnu))) return rc;
    for (int64_t i = 0; i < nnu; i++)
        if (!(nu[i] >= 0)) return fail(CS_EINVAL, "wavenumbers must be positive");
    const int M = nT * nP;
    std::vector<double> Ts(M), Ps(M), Pp(M);
    for (int j = 0; j < nP; j++)
        for (int i = 0; i < nT; i++) {
            const double C = conc[i + (size_t)nT * j];
            if (!(C >= 0 && C <= 1)) return fail(CS_EINVAL, "gas molar concentrations must be in [0,1], not %g (encountered @ %g K, %g Pa)", C, T[i], P[j]);
            Ts[i + nT * j] = T[i];
            Ps[i + nT * j] = P[j];
            Pp[i + nT * j] = C * P[j];
        }
    GasTable &G = ctx->gas[gas_slot];
    if ((rc = check_gas_states(G, M, Ts.data()))) return rc;
    if (shape == SH_VOIGT && (rc = check_near_density(G, nu, nnu, dnu_cut))) return rc;
    ph_set_grid(ctx, ctx->ph, nu, nnu, ++g_grid_counter);
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    TableDev &tb = ctx->tab[table_slot];
    tb.present = false;
    int64_t g0, g1, pairs, inr;
    included_range(G.h_nu, nu[0], nu[nnu - 1], dnu_cut, true, g0, g1);
    std::vector<int32_t> J0, J1;
    tile_windows(G.h_nu, g0, g1, nu, nnu, window_reach(shape, G, nu[nnu - 1], dnu_cut), J0, J1, pairs, inr);
    std::vector<WaveWin> win;
    const int xtiles = wave_windows(G.h_nu, g0, g1, nu, nnu, dnu_cut, win);
    const int ntile = (int)J0.size();
    DevBuf dnu, dT, dP, dPp, dJ0, dJ1, hot, cold, dwin, dzones, dgb, dranges, dlrt, dqref;
    {
        std::vector<double> lrt, qr;
        state_tables(G, M, Ts.data(), lrt, qr);
        if ((rc = upload(dlrt, lrt.data(), lrt.size(), s)) || (rc = upload(dqref, qr.data(), qr.size(), s))) return rc;
        HIPCHK(hipStreamSynchronize(s));
    }
    std::vector<double> gb = gamma_bound(G, M, Ts.data(), Ps.data(), Pp.data());
    if ((rc = upload(dnu, nu, nnu, s)) || (rc = upload(dT, Ts.data(), M, s)) || (rc = upload(dP, Ps.data(), M, s)) ||
        (rc = upload(dPp, Pp.data(), M, s)) || (rc = upload(dJ0, J0.data(), ntile, s)) || (rc = upload(dJ1, J1.data(), ntile, s)) ||
        (rc = upload(dwin, win.data(), win.size(), s)) || (rc = upload(dgb, gb.data(), M, s)))
        return rc;
    HIPCHK(tb.Z.reserve((size_t)M * nnu * sizeof(double)));
    const size_t per_state = (size_t)G.L * (sizeof(LineHot) + sizeof(LineCold)) + (size_t)nnu * sizeof(int2);
    const int kc = (int)std::max<size_t>(1, std::min<size_t>({(size_t)M, ((size_t)4 << 30) / per_state, (size_t)65535}));
    HIPCHK(hot.reserve(((size_t)kc * G.L + 4) * sizeof(LineHot)));
    HIPCHK(cold.reserve((size_t)kc * G.L * sizeof(LineCold)));
    HIPCHK(dzones.reserve((size_t)kc * win.size() * sizeof(Zone)));
    HIPCHK(dranges.reserve((size_t)kc * nnu * sizeof(int2) + (size_t)2 * kc * ((nnu + 63) / 64) * sizeof(int)));   // + per-(tile, state) flags
    LineF32 *mix32 = nullptr;
    if (ctx->mixed && shape == SH_VOIGT) {
        HIPCHK(ctx->hot32.reserve(((size_t)kc * G.L + 4) * sizeof(LineF32)));
        mix32 = ctx->hot32.as<LineF32>();
    }
    ChebGrid cheb;
    GasInterp ginterp;
    Interp itp;
    if (ctx->interp && (shape == SH_VOIGT || shape == SH_LORENTZ)) {
        if ((rc = cheb_build(ctx, cheb, nu, dnu.as<double>(), nnu, dnu_cut, s)) ||
            (rc = gas_interp_build(ctx, ginterp, cheb, G.h_nu, g0, g1, nu, nnu, dnu_cut, kc, s)))
            return rc;
        itp = interp_view(cheb, ginterp, kc);
        if (!ctx->matrix_nodes) itp.sep = nullptr, itp.edge = nullptr;
        itp.sep_always = ctx->matrix_nodes == 2;
        interp_settings(ctx, itp);
        itp.core = ctx->matrix_core != 0;
    }
    for (int k0 = 0; k0 < M; k0 += kc) {
        const int kn = std::min(kc, M - k0);
        launch_gas(s, shape, G, J0.front(), J1.back(), kn, dT.as<double>() + k0, dP.as<double>() + k0, dPp.as<double>() + k0, nullptr, 0, dlrt.as<double>() + k0,
                   dqref.as<double>() + (size_t)k0 * G.niso, hot.as<LineHot>(), cold.as<LineCold>(), dnu.as<double>(), nnu, ntile, dJ0.as<int32_t>(), dJ1.as<int32_t>(), dwin.as<WaveWin>(), xtiles,
                   dzones.as<Zone>(), dranges.as<int2>(), dgb.as<double>() + k0, dnu_cut, 0.0, nullptr, tb.Z.as<double>() + (size_t)k0 * nnu, 0, nullptr,
                   mix32, ctx->far_s, itp, nullptr, &ctx->ph);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
    }
    CS_LAUNCH(k_table_log, dim3((unsigned)((nnu + 255) / 256)), dim3(256), 0, s, tb.Z.as<double>(), M, nnu);
    HIPCHK(hipGetLastError());
    if (lnsigma_out) HIPCHK(hipMemcpyAsync(lnsigma_out, tb.Z.p, (size_t)M * nnu * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    tb.nnu = nnu; tb.nT = nT; tb.nP = nP;
    tb.T.assign(T, T + nT);
    tb.lnP.resize(nP);
    for (int j = 0; j < nP; j++) tb.lnP[j] = std::log(P[j]);
    tb.nu.assign(nu, nu + nnu);
    tb.generation = next_generation();
    tb.present = true;
    return CS_OK;
}

int cs_table_clear(cs_ctx *ctx, int table_slot)
{
    if (!ctx || table_slot < 0 || table_slot >= CS_MAX_TABLE) return fail(CS_EINVAL, "bad table slot");
    ctx->tab[table_slot] = TableDev();
    return CS_OK;
}

// sigma[k][nu] += conc[k] * exp(sum_m Z[m][nu] W[m][k]) for K states (the Gas functor, gases.jl:85,278)
#ifndef CS_TABLE_NSUB
#define CS_TABLE_NSUB 2
#endif
static int launch_table_eval(hipStream_t s, const double *Z, int M, int64_t nnu, const double *W, int K, const double *conc, double *sigma)
{
    const int nt64 = (int)((nnu + 63) / 64);
    const int nst = (K + 15) / 16, nsg = (nst + CS_TABLE_NSUB - 1) / CS_TABLE_NSUB;
    const int64_t nblk = (int64_t)((nt64 + 3) / 4) * nsg;
    if (nblk > 0x7fffffffLL) return fail(CS_EINVAL, "too many (tile, state) blocks for the opacity-table kernel");
    CS_LAUNCH(k_table_eval_mfma<CS_TABLE_NSUB>, dim3((unsigned)nblk), dim3(256), 0, s, Z, M, nnu, nt64, W, K, conc, sigma);
    return CS_OK;
}

// W[m][k] = a_i(T_k) b_j(ln P_k), m = i + nT*j
static int table_weights(const TableDev &tb, int K, const double *Tk, const double *Pk, std::vector<double> &W)
{
    const int M = tb.nT * tb.nP;
    W.assign((size_t)M * K, 0.0);
    std::vector<double> a, b;
    for (int k = 0; k < K; k++) {
        const double lp = std::log(Pk[k]);
        if (!(Tk[k] >= tb.T.front() && Tk[k] <= tb.T.back()))
            return fail(CS_EINVAL, "temperature %g K outside the opacity table's domain [%g, %g]", Tk[k], tb.T.front(), tb.T.back());
        if (!(lp >= tb.lnP.front() - 1e-12 && lp <= tb.lnP.back() + 1e-12))
            return fail(CS_EINVAL, "Pressure %g Pa outside the opacity table's domain [%g, %g]", Pk[k], std::exp(tb.lnP.front()), std::exp(tb.lnP.back()));
        cheb_basis(tb.T, Tk[k], a);
        cheb_basis(tb.lnP, std::min(std::max(lp, tb.lnP.front()), tb.lnP.back()), b);
        for (int j = 0; j < tb.nP; j++)
            for (int i = 0; i < tb.nT; i++) W[(size_t)(i + tb.nT * j) * K + k] = a[i] * b[j];
    }
    return CS_OK;
}

int cs_table_eval(cs_ctx *ctx, int table_slot, double T, double P, int64_t i0, int64_t n, double *sigma_out)
{
    if (!ctx || table_slot < 0 || table_slot >= CS_MAX_TABLE || !ctx->tab[table_slot].present) return fail(CS_EINVAL, "table slot is empty");
    TableDev &tb = ctx->tab[table_slot];
    if (i0 < 0 || n < 1 || i0 + n > tb.nnu) return fail(CS_EINVAL, "wavenumber range out of bounds");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<double> W;
    int rc;
    if ((rc = table_weights(tb, 1, &T, &P, W))) return rc;
    const int M = tb.nT * tb.nP;
    const double one = 1.0;
    if ((rc = upload(ctx->tmpA, W.data(), M, s)) || (rc = upload(ctx->tmpB, &one, 1, s))) return rc;
    HIPCHK(ctx->tmpC.reserve((size_t)tb.nnu * sizeof(double)));
    HIPCHK(hipMemsetAsync(ctx->tmpC.p, 0, (size_t)tb.nnu * sizeof(double), s));
    if ((rc = launch_table_eval(s, tb.Z.as<double>(), M, tb.nnu, ctx->tmpA.as<double>(), 1, ctx->tmpB.as<double>(), ctx->tmpC.as<double>()))) return rc;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sigma_out, ctx->tmpC.as<double>() + i0, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return CS_OK;
}

static int upload_tables(cs_ctx *ctx, const double *conc_tab)
{
    Column &c = ctx->col;
    hipStream_t s = ctx->stream;
    const int nt = (int)c.tab.size();
    std::vector<double> W, cc(c.K);
    int rc;
    for (int t = 0; t < nt; t++) {
        TableDev &tb = ctx->tab[c.tab[t].slot];
        if ((rc = table_weights(tb, c.K, c.h_Tk.data(), c.h_Pk.data(), W))) return rc;
        for (int k = 0; k < c.K; k++) {
            cc[k] = conc_tab[t + (size_t)nt * k];
            if (!(cc[k] >= 0 && cc[k] <= 1)) return fail(CS_EINVAL, "gas molar concentrations must be in [0,1], not %g", cc[k]);
        }
        if ((rc = upload(c.tab[t].W, W.data(), W.size(), s)) || (rc = upload(c.tab[t].conc, cc.data(), c.K, s))) return rc;
    }
    HIPCHK(hipStreamSynchronize(s));
    return CS_OK;
}

int cs_column_set_tables(cs_ctx *ctx, int ntab, const int *table_slots, const double *conc_tab)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "cs_column_setup has not been called");
    if (ntab < 0 || ntab > CS_MAX_TABLE) return fail(CS_EINVAL, "ntab out of range");
    Column &c = ctx->col;
    HIPCHK(hipSetDevice(ctx->device));
    drop_graph(c);
    c.tab.clear();
    c.tab.resize(ntab);
    for (int t = 0; t < ntab; t++) {
        const int sl = table_slots[t];
        if (sl < 0 || sl >= CS_MAX_TABLE || !ctx->tab[sl].present) return fail(CS_EINVAL, "table slot %d is empty", sl);
        TableDev &tb = ctx->tab[sl];
        if (tb.nnu != c.nnu || !std::equal(tb.nu.begin(), tb.nu.end(), c.h_nu.begin()))
            return fail(CS_EINVAL, "gases must have identical wavenumber vectors");
        c.tab[t].slot = sl;
        c.tab[t].generation = tb.generation;
    }
    int rc = upload_tables(ctx, conc_tab);
    if (rc) c.tab.clear();
    return rc;
}

int cs_cia_begin(cs_ctx *ctx, int cia_slot, int nband)
{
    if (!ctx || cia_slot < 0 || cia_slot >= CS_MAX_CIA) return fail(CS_EINVAL, "bad CIA slot");
    if (nband < 1 || nband > CS_MAX_CIA_BAND) return fail(CS_EINVAL, "a CIA object holds 1..%d bands", CS_MAX_CIA_BAND);
    ctx->cia[cia_slot] = CiaDev();
    ctx->cia[cia_slot].bands.resize(nband);
    ctx->cia[cia_slot].generation = next_generation();
    return CS_OK;
}

int cs_cia_band(cs_ctx *ctx, int cia_slot, int band, int nb, const double *nu_b, int nt, const double *T_b, const double *lnk)
{
    if (!ctx || cia_slot < 0 || cia_slot >= CS_MAX_CIA) return fail(CS_EINVAL, "bad CIA slot");
    CiaDev &cd = ctx->cia[cia_slot];
    if (band < 0 || band >= (int)cd.bands.size()) return fail(CS_EINVAL, "band index out of range");
    if (nb < 2 || nt < 1) return fail(CS_EINVAL, "a band needs at least 2 wavenumbers and 1 temperature");
    for (int i = 1; i < nb; i++)
        if (!(nu_b[i] > nu_b[i - 1])) return fail(CS_EORDER, "band wavenumbers must be ascending");
    for (int j = 1; j < nt; j++)
        if (!(T_b[j] > T_b[j - 1])) return fail(CS_EORDER, "band temperatures must be ascending");
    HIPCHK(hipSetDevice(ctx->device));
    CiaBandHost &b = cd.bands[band];
    b.nu.assign(nu_b, nu_b + nb);
    b.T.assign(T_b, T_b + nt);
    int rc;
    if ((rc = upload(b.dnu, nu_b, nb, ctx->stream)) || (rc = upload(b.dlnk, lnk, (size_t)nb * nt, ctx->stream))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    cd.filled++;
    cd.present = cd.filled >= (int)cd.bands.size();
    cd.generation = next_generation();
    return CS_OK;
}

int cs_cia_clear(cs_ctx *ctx, int cia_slot)
{
    if (!ctx || cia_slot < 0 || cia_slot >= CS_MAX_CIA) return fail(CS_EINVAL, "bad CIA slot");
    ctx->cia[cia_slot] = CiaDev();
    return CS_OK;
}

// per-state inputs of a CIA pair: temperature cells of every band + number densities (cia(k,T,Pa,P1,P2), :295-303).
// Kn states with temperatures T, air pressures Pa and partial pressures P1/P2 (element k at P[idx + stride*k]); results go to the
// given device buffers (a column's own, or a batch's)
static int upload_cia_states(cs_ctx *ctx, int slot, int flags, int Kn, const double *T, const double *Pa, const double *P1,
                             const double *P2, int stride, int idx, DevBuf &dst, DevBuf &d1, DevBuf &d2, DevBuf &da)
{
    CiaDev &cd = ctx->cia[slot];
    const int K = Kn, nband = (int)cd.bands.size();
    const bool extrap = flags & 1, singles = flags & 2;
    std::vector<CiaState> st((size_t)nband * K);
    for (int b = 0; b < nband; b++) {
        const std::vector<double> &Tg = cd.bands[b].T;
        const int nt = (int)Tg.size();
        for (int k = 0; k < K; k++) {
            CiaState s;
            s.use = 0; s.jT = 0; s.fT = 0.0;
            const double Tk = T[k];
            if (nt == 1) {
                s.use = singles ? 1 : 0;
            } else {
                double Te = Tk;
                if (Tk >= Tg.front() && Tk <= Tg.back()) s.use = 1;                       // :258
                else if (extrap) { s.use = 1; Te = Tk > Tg.back() ? Tg.back() : Tg.front(); }  // :261-263
                if (s.use) {
                    int j = (int)(std::upper_bound(Tg.begin(), Tg.end(), Te) - Tg.begin()) - 1;
                    j = std::min(std::max(j, 0), nt - 2);
                    s.jT = j;
                    s.fT = (Te - Tg[j]) / (Tg[j + 1] - Tg[j]);
                }
            }
            st[(size_t)b * K + k] = s;
        }
    }
    std::vector<double> r1(K), r2(K), ra(K);
    for (int k = 0; k < K; k++) {
        r1[k] = (P1[idx + (size_t)stride * k] / kAtm) * (273.15 / T[k]);   // amagat, :297-298 (T0 = 273.15, constants.jl:23)
        r2[k] = (P2[idx + (size_t)stride * k] / kAtm) * (273.15 / T[k]);
        ra[k] = 1e-6 * Pa[k] / (kKb * T[k]);                               // molecules/cm^3, :300
    }
    hipStream_t s = ctx->stream;
    int rc;
    if ((rc = upload(dst, st.data(), st.size(), s)) || (rc = upload(d1, r1.data(), K, s)) ||
        (rc = upload(d2, r2.data(), K, s)) || (rc = upload(da, ra.data(), K, s)))
        return rc;
    HIPCHK(hipStreamSynchronize(s));   // the host vectors are locals
    return CS_OK;
}
static int upload_cia_state(cs_ctx *ctx, ColCia &cc, const double *P1, const double *P2, int stride, int idx)
{
    Column &c = ctx->col;
    return upload_cia_states(ctx, cc.slot, cc.flags, c.K, c.h_Tk.data(), c.h_Pk.data(), P1, P2, stride, idx, cc.st, cc.rho1, cc.rho2, cc.rhoa);
}

int cs_column_set_cia(cs_ctx *ctx, int ncia, const int *cia_slots, const int *flags, const double *P1, const double *P2)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "cs_column_setup has not been called");
    if (ncia < 0 || ncia > CS_MAX_CIA) return fail(CS_EINVAL, "ncia out of range");
    Column &c = ctx->col;
    HIPCHK(hipSetDevice(ctx->device));
    for (int t = 0; t < ncia; t++)
        if (cia_slots[t] < 0 || cia_slots[t] >= CS_MAX_CIA || !ctx->cia[cia_slots[t]].present) {
            c.cia.clear();
            return fail(CS_EINVAL, "CIA slot %d is empty", cia_slots[t]);
        }
    // what depends on the grid and the tables alone (band descriptors, sample cells of every wavenumber) is kept while the pairs named
    // are the ones the column already holds: an RCM loop re-sets its pairs on every call for the partial pressures only
    bool same = (int)c.cia.size() == ncia;
    for (int t = 0; t < ncia && same; t++)
        same = c.cia[t].slot == cia_slots[t] && c.cia[t].generation == ctx->cia[cia_slots[t]].generation;
    if (!same) {
        drop_graph(c);
        c.cia.clear();
        c.cia.resize(ncia);
    }
    int rc;
    const int nt64 = (int)((c.nnu + 63) / 64);
    for (int t = 0; t < ncia; t++) {
        ColCia &cc = c.cia[t];
        cc.slot = cia_slots[t];
        cc.flags = flags ? flags[t] : 0;
        CiaDev &cd = ctx->cia[cc.slot];
        if (!same) {
            cc.generation = cd.generation;
            cc.nband = (int)cd.bands.size();
            std::vector<CiaBand> hb(cc.nband);
            for (int b = 0; b < cc.nband; b++) {
                hb[b].nu = cd.bands[b].dnu.as<double>();
                hb[b].lnk = cd.bands[b].dlnk.as<double>();
                hb[b].nb = (int)cd.bands[b].nu.size();
                hb[b].nt = (int)cd.bands[b].T.size();
            }
            // k_flux: room for ln k at every state's temperature; per tile the bands that reach it; per wavenumber its cell in each
            std::vector<int64_t> toff(cc.nband);
            int64_t tot = 0;
            for (int b = 0; b < cc.nband; b++) { toff[b] = tot; tot += (int64_t)c.K * hb[b].nb; }
            std::vector<int32_t> tband((size_t)nt64 * CS_CIA_ACT, -1);
            cc.max_overlap = 0;
            for (int ti = 0; ti < nt64; ti++) {
                const double vlo = c.h_nu[(int64_t)ti * 64], vhi = c.h_nu[std::min<int64_t>((int64_t)ti * 64 + 63, c.nnu - 1)];
                int n = 0;
                for (int b = 0; b < cc.nband; b++)
                    if (cd.bands[b].nu.front() <= vhi && cd.bands[b].nu.back() >= vlo) {
                        if (n < CS_CIA_ACT) tband[(size_t)ti * CS_CIA_ACT + n] = b;
                        n++;
                    }
                cc.max_overlap = std::max(cc.max_overlap, n);
            }
            const int nslot = std::min(std::max(cc.max_overlap, 1), (int)CS_CIA_ACT);
            std::vector<int32_t> cell((size_t)nslot * c.nnu, -1);
            std::vector<double> fx((size_t)nslot * c.nnu, 0.0);
            for (int ti = 0; ti < nt64; ti++)
                for (int q = 0; q < nslot; q++) {
                    const int b = tband[(size_t)ti * CS_CIA_ACT + q];
                    if (b < 0) continue;
                    const std::vector<double> &g = cd.bands[b].nu;
                    for (int64_t i = (int64_t)ti * 64; i < std::min<int64_t>((int64_t)ti * 64 + 64, c.nnu); i++) {
                        const double v = c.h_nu[i];
                        if (!(g.front() <= v && v <= g.back())) continue;      // Phi.G.xa <= nu <= Phi.G.xb, :255
                        int lo = 0, hi = (int)g.size() - 1;                      // cell with g[lo] <= v < g[lo + 1] (the last one for v == g.back()): k_cia's search
                        while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (g[m] <= v) lo = m; else hi = m; }
                        cell[(size_t)q * c.nnu + i] = lo;
                        fx[(size_t)q * c.nnu + i] = (v - g[lo]) / (g[lo + 1] - g[lo]);
                    }
                }
            if ((rc = upload(cc.bands, hb.data(), hb.size(), ctx->stream)) || (rc = upload(cc.toff, toff.data(), toff.size(), ctx->stream)) ||
                (rc = upload(cc.tband, tband.data(), tband.size(), ctx->stream)) || (rc = upload(cc.cell, cell.data(), cell.size(), ctx->stream)) ||
                (rc = upload(cc.fx, fx.data(), fx.size(), ctx->stream))) {
                c.cia.clear();
                return rc;
            }
            HIPCHK(cc.tab.reserve((size_t)tot * sizeof(double)));
            HIPCHK(cc.fac.reserve((size_t)c.K * sizeof(double)));
            HIPCHK(hipStreamSynchronize(ctx->stream));   // (the host vectors are locals)
        }
        if ((rc = upload_cia_state(ctx, cc, P1, P2, ncia, t))) { c.cia.clear(); return rc; }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

static int sigma_impl(cs_ctx *ctx, hipStream_t s, hipEvent_t *ev, int &e, bool *near_plane_live = nullptr, FluxFuse *fuse = nullptr);
static int column_current(cs_ctx *ctx);

// ---- AcceleratedAbsorber (absorbers.jl:114-203) -------------------------------------------------------------------------
int cs_accel_store(cs_ctx *ctx, int accel_slot)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "cs_column_setup has not been called");
    if (accel_slot < 0 || accel_slot >= CS_MAX_ACCEL) return fail(CS_EINVAL, "accelerated-absorber slot %d out of range", accel_slot);
    Column &c = ctx->col;
    if (c.accel.slot >= 0) return fail(CS_ESTATE, "the resident column is itself an accelerated absorber");
    if (c.K < 2) return fail(CS_EINVAL, "need at least two pressure knots");
    for (int k = 1; k < c.K; k++)
        if (!(c.h_Pk[k] > c.h_Pk[k - 1])) return fail(CS_EORDER, "knot pressures must be strictly ascending");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int e = 0, rc;
    if ((rc = sigma_impl(ctx, s, nullptr, e))) return rc;   // Sigma(U, i, T_k, P_k) for every i and knot k (update!, absorbers.jl:173-200)
    AccelDev &ad = ctx->accel[accel_slot];
    const int64_t n = (int64_t)c.K * c.nnu;
    HIPCHK(ad.L.reserve((size_t)n * sizeof(double)));
    CS_LAUNCH(k_accel_store, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, c.sigma.as<double>(), ad.L.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    std::vector<double> lnP(c.K);
    for (int k = 0; k < c.K; k++) lnP[k] = std::log(c.h_Pk[k]);
    if (!ad.present || ad.nnu != c.nnu || ad.nk != c.K || ad.lnP != lnP || ad.nu != c.h_nu) ad.generation = next_generation();   // other knots or grid (new values on the same ones: update!, absorbers.jl:173-200)
    ad.nnu = c.nnu;
    ad.nk = c.K;
    ad.nu = c.h_nu;
    ad.lnP = lnP;
    ad.present = true;
    return CS_OK;
}

int cs_accel_clear(cs_ctx *ctx, int accel_slot)
{
    if (!ctx || accel_slot < 0 || accel_slot >= CS_MAX_ACCEL) return fail(CS_EINVAL, "bad accelerated-absorber slot");
    if (ctx->col.accel.slot == accel_slot) { ctx->col.accel.slot = -1; ctx->col.ready = false; }
    ctx->accel[accel_slot] = AccelDev();
    return CS_OK;
}

// knot interval and abscissae of LinearInterpolator(lnP, y, NoBoundaries())(x): the cell of x clamped to the end cells (extrapolation)
static void accel_cells(const AccelDev &ad, int K, const double *P, std::vector<int32_t> &cell, std::vector<double> &x,
                        std::vector<double> &xa, std::vector<double> &xb)
{
    cell.resize(K); x.resize(K); xa.resize(K); xb.resize(K);
    for (int k = 0; k < K; k++) {
        x[k] = std::log(P[k]);
        int i = (int)(std::upper_bound(ad.lnP.begin(), ad.lnP.end(), x[k]) - ad.lnP.begin()) - 1;
        i = std::min(std::max(i, 0), ad.nk - 2);
        cell[k] = i;
        xa[k] = ad.lnP[i];
        xb[k] = ad.lnP[i + 1];
    }
}

int cs_accel_eval(cs_ctx *ctx, int accel_slot, double P, int64_t i0, int64_t n, double *sigma_out)
{
    if (!ctx || accel_slot < 0 || accel_slot >= CS_MAX_ACCEL || !ctx->accel[accel_slot].present) return fail(CS_EINVAL, "accelerated-absorber slot is empty");
    AccelDev &ad = ctx->accel[accel_slot];
    if (i0 < 0 || n < 1 || i0 + n > ad.nnu || !sigma_out) return fail(CS_EINVAL, "wavenumber range out of bounds");
    if (!(P > 0)) return fail(CS_EINVAL, "pressure must be positive");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    std::vector<int32_t> cell;
    std::vector<double> x, xa, xb;
    accel_cells(ad, 1, &P, cell, x, xa, xb);
    DevBuf dc, dx, da, db;
    int rc;
    if ((rc = upload(dc, cell.data(), 1, s)) || (rc = upload(dx, x.data(), 1, s)) || (rc = upload(da, xa.data(), 1, s)) ||
        (rc = upload(db, xb.data(), 1, s)))
        return rc;
    HIPCHK(ctx->tmpC.reserve((size_t)ad.nnu * sizeof(double)));
    CS_LAUNCH(k_accel_eval, dim3((unsigned)((ad.nnu + 255) / 256), 1), dim3(256), 0, s, ad.L.as<double>(), ad.nnu, 1, dc.as<int32_t>(),
                       dx.as<double>(), da.as<double>(), db.as<double>(), 0.0, (const double *)nullptr, ctx->tmpC.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sigma_out, ctx->tmpC.as<double>() + i0, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return CS_OK;
}

int cs_column_set_accel(cs_ctx *ctx, int accel_slot)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "cs_column_setup has not been called");
    Column &c = ctx->col;
    drop_graph(c);
    if (accel_slot < 0) { c.accel.slot = -1; return CS_OK; }
    if (accel_slot >= CS_MAX_ACCEL || !ctx->accel[accel_slot].present) return fail(CS_EINVAL, "accelerated-absorber slot %d is empty", accel_slot);
    // an AcceleratedAbsorber stands for ALL absorbers of a column (unifyabsorbers(::Tuple{AcceleratedAbsorber}), absorbers.jl:216)
    if (c.ngas > 0 || !c.tab.empty() || !c.cia.empty()) return fail(CS_EINVAL, "a column over an accelerated absorber has no other absorbers");
    AccelDev &ad = ctx->accel[accel_slot];
    if (ad.nnu != c.nnu || !std::equal(ad.nu.begin(), ad.nu.end(), c.h_nu.begin())) return fail(CS_EINVAL, "wavenumber grids differ");
    HIPCHK(hipSetDevice(ctx->device));
    std::vector<int32_t> cell;
    std::vector<double> x, xa, xb;
    accel_cells(ad, c.K, c.h_Pk.data(), cell, x, xa, xb);
    int rc;
    hipStream_t s = ctx->stream;
    if ((rc = upload(c.accel.cell, cell.data(), c.K, s)) || (rc = upload(c.accel.x, x.data(), c.K, s)) ||
        (rc = upload(c.accel.xa, xa.data(), c.K, s)) || (rc = upload(c.accel.xb, xb.data(), c.K, s)))
        return rc;
    HIPCHK(hipStreamSynchronize(s));
    c.accel.slot = accel_slot;
    c.accel.generation = ad.generation;
    return CS_OK;
}

int cs_column_setup(cs_ctx *ctx, int64_t nnu, const double *nu, const double *wts, int np, const double *P, double g,
                    int nlobatto, const double *T_nodes, const double *mu_nodes, const double *T_levels, int ngas,
                    const int *gas_slots, const int *shapes, const double *dnu_cuts, const double *conc,
                    double sigma_gray, const double *sigma_extra, const double *S_toa, const double *albedo,
                    double theta_s, int nstream, int want_tau, int want_M)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    Column &c = ctx->col;
    c.ready = false;
    c.F_dst = nullptr;      // (a destination for the band fluxes belongs to the column it was set for)
    drop_graph(c);
    if (np < 2) return fail(CS_EINVAL, "need at least two pressure levels");
    if (nlobatto < 2 || nlobatto > CS_MAX_LOBATTO) return fail(CS_EINVAL, "nlobatto must be in [2,%d]", CS_MAX_LOBATTO);
    if (nstream < 1 || nstream > CS_MAX_STREAM) return fail(CS_EINVAL, "nstream must be in [1,%d]", CS_MAX_STREAM);
    if (ngas < 0 || ngas > CS_MAX_GAS) return fail(CS_EINVAL, "ngas out of range");
    if (!(theta_s >= 0 && theta_s < M_PI / 2)) return fail(CS_EINVAL, "azimuth angle theta must be in [0,pi/2)");
    if (!(g > 0)) return fail(CS_EINVAL, "g must be positive");
    if ((int64_t)(np - 1) * (nlobatto - 1) + 1 > 65535) return fail(CS_EINVAL, "too many node states for one launch (65535)");
    int rc;
    if ((rc = check_ascending(nu, nnu))) return rc;
    for (int i = 1; i < np; i++)
        if (!(P[i] >= P[i - 1])) return fail(CS_EORDER, "pressure coordinates must be in ascending order (sorted)");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    c.nnu = nnu; c.np = np; c.nl = np - 1; c.nlob = nlobatto; c.K = (np - 1) * (nlobatto - 1) + 1;
    c.nstream = nstream; c.ngas = ngas; c.ntile = (int)((nnu + 255) / 256);
    c.rtg = rt_geometry(nnu, np, 1, nstream, ctx->tune[5] != 0);
    c.want_tau = want_tau != 0; c.want_M = want_M != 0;
    c.g = g; c.sigma_gray = sigma_gray; c.theta_s = theta_s;
    const int K = c.K, nl = c.nl;
    // quadrature rules
    RtParams &rt = c.rt;
    memset(&rt, 0, sizeof rt);
    rt.np = np; rt.nlobatto = nlobatto; rt.K = K; rt.nstream = nstream;
    rt.C = 1e-4 * kNa / g;
    rt.cos_ts = std::cos(theta_s);
    c.h_xs.assign(nlobatto, 0.0);
    cs_lobattonodes(nlobatto, c.h_xs.data(), rt.ws);
    cs_streamnodes(nstream, rt.m, rt.W);
    for (int k = 0; k < nstream; k++) rt.im[k] = 1.0 / rt.m[k];
    // node states: k = i*(nlobatto-1) + n   (discretized.jl:150,162,169)
    c.h_P.assign(P, P + np);
    c.h_nu.assign(nu, nu + nnu);
    c.h_Pk.assign(K, 0.0);
    c.h_Pk[0] = P[0];
    for (int i = 0; i < nl; i++) {
        const double dP = P[i + 1] - P[i];
        for (int n = 1; n < nlobatto; n++)
            c.h_Pk[i * (nlobatto - 1) + n] = (n == nlobatto - 1) ? P[i + 1] : P[i] + dP * c.h_xs[n];
    }
    std::vector<double> wt(nnu);
    if (wts) {
        std::copy(wts, wts + nnu, wt.begin());
    } else {  // trapz (util.jl:26-33) as per-point weights
        for (int64_t j = 0; j < nnu; j++) {
            double a = j > 0 ? nu[j] - nu[j - 1] : 0.0, b = j + 1 < nnu ? nu[j + 1] - nu[j] : 0.0;
            wt[j] = (a + b) / 2;
        }
    }
    if ((rc = upload(c.nu, nu, nnu, s)) || (rc = upload(c.wts, wt.data(), nnu, s)) || (rc = upload(c.P, P, np, s)) ||
        (rc = upload(c.Pk, c.h_Pk.data(), K, s)))
        return rc;
    c.default_wts = wts == nullptr;
    c.g_nnu = 0; c.g_start = 0; c.g_left = c.g_right = 0.0;   // (cs_fluxes_discretized_multi marks its shards after a successful setup)
    c.interp = interp_key(ctx);   // every setting of the context that shapes what setup builds
    c.has_extra = sigma_extra != nullptr;
    // an all-zero stellar spectrum / albedo is the same as none (0*exp(..) and M*0/pi are exact zeros): skip their work, and let
    // the upward sweep start without waiting for the downward one (k_rt<.., UD>)
    c.has_S = S_toa != nullptr && std::any_of(S_toa, S_toa + nnu, [](double x) { return x != 0.0; });
    c.has_alb = albedo != nullptr && std::any_of(albedo, albedo + nnu, [](double x) { return x != 0.0; });
    if (c.has_extra && (rc = upload(c.extra, sigma_extra, (size_t)nnu * K, s))) return rc;
    if (c.has_S && (rc = upload(c.S_toa, S_toa, nnu, s))) return rc;
    if (c.has_alb && (rc = upload(c.albedo, albedo, nnu, s))) return rc;
    // gases: launch groups, tile windows and workspace
    c.gas.clear();
    c.ugas.clear();
    c.ugas.resize(ngas);
    c.merge = ctx->merge;
    c.grid_id = ++g_grid_counter;
    c.cheb.nlev = 0;
    if (ctx->interp) {   // interval sizes from the narrowest Voigt cut-off of the column
        double cmin = 0.0;
        for (int gi = 0; gi < ngas; gi++)
            if ((shapes ? shapes[gi] : CS_SHAPE_VOIGT) == SH_VOIGT || (shapes ? shapes[gi] : CS_SHAPE_VOIGT) == SH_LORENTZ) {
                const double cu = dnu_cuts ? dnu_cuts[gi] : 25.0;
                cmin = cmin > 0.0 ? std::min(cmin, cu) : cu;
            }
        if (cmin > 0.0 && (rc = cheb_build(ctx, c.cheb, nu, c.nu.as<double>(), nnu, cmin, s))) return rc;
    }
    std::vector<std::vector<int>> groups;
    for (int gi = 0; gi < ngas; gi++) {
        UserGas &ug = c.ugas[gi];
        ug.slot = gas_slots[gi];
        ug.shape = shapes ? shapes[gi] : CS_SHAPE_VOIGT;
        ug.cut = dnu_cuts ? dnu_cuts[gi] : 25.0;
        if (ug.slot < 0 || ug.slot >= CS_MAX_GAS || !ctx->gas[ug.slot].present)
            return fail(CS_EINVAL, "gas slot %d is empty", ug.slot);
        ug.generation = ctx->gas[ug.slot].generation;
        if (ug.shape < 0 || ug.shape > 3) return fail(CS_EINVAL, "unknown shape %d", ug.shape);
        const GasTable &G = ctx->gas[ug.slot];
        ug.pairs_per_state = -1;   // counted on demand (cs_column_counts): O(nnu log L) on the host
        ug.lines_in_range = std::upper_bound(G.h_nu.begin(), G.h_nu.end(), nu[nnu - 1] + ug.cut) -
                            std::lower_bound(G.h_nu.begin(), G.h_nu.end(), nu[0] - ug.cut);   // (the reference's count: inside the cut-off)
        // Voigt (Lorentz) gases with the same cut-off go into one group; a slot named twice stays apart (a merged table tags a
        // line with ONE member)
        bool placed = false;
        if (ctx->merge && (ug.shape == SH_VOIGT || ug.shape == SH_LORENTZ))
            for (auto &grp : groups) {
                const UserGas &h = c.ugas[grp[0]];
                bool dup = false;
                for (int m : grp) dup = dup || c.ugas[m].slot == ug.slot;
                if (h.shape == ug.shape && h.cut == ug.cut && !dup && grp.size() < 255) { grp.push_back(gi); placed = true; break; }
            }
        if (!placed) groups.push_back(std::vector<int>{gi});
    }
    c.gas.resize(groups.size());
    size_t maxL = 0;
    for (size_t qi = 0; qi < groups.size(); qi++) {
        ColGas &cg = c.gas[qi];
        cg.mem = groups[qi];
        cg.shape = c.ugas[cg.mem[0]].shape;
        cg.cut = c.ugas[cg.mem[0]].cut;
        if (cg.mem.size() == 1) {
            cg.tab = &ctx->gas[c.ugas[cg.mem[0]].slot];
        } else {
            std::vector<int> slots;
            for (int m : cg.mem) slots.push_back(c.ugas[m].slot);
            if ((rc = merged_table(ctx, slots, &cg.hold))) return rc;
            cg.tab = cg.hold.get();
        }
        const GasTable &G = *cg.tab;
        if (cg.shape == SH_VOIGT && (rc = check_near_density(G, nu, nnu, cg.cut))) return rc;
        int64_t g0, g1, pairs_unused, inr_unused;
        included_range(G.h_nu, nu[0], nu[nnu - 1], cg.cut, false, g0, g1);
        std::vector<int32_t> J0, J1;
        tile_windows(G.h_nu, g0, g1, nu, nnu, window_reach(cg.shape, G, nu[nnu - 1], cg.cut), J0, J1, pairs_unused, inr_unused);
        cg.jlo = J0.front();
        cg.jhi = J1.back();
        std::vector<WaveWin> win;
        cg.xtiles = wave_windows(G.h_nu, g0, g1, nu, nnu, cg.cut, win);
        if ((rc = upload(cg.J0, J0.data(), J0.size(), s)) || (rc = upload(cg.J1, J1.data(), J1.size(), s)) ||
            (rc = upload(cg.win, win.data(), win.size(), s)))
            return rc;
        HIPCHK(cg.zones.reserve((size_t)c.K * win.size() * sizeof(Zone)));
        HIPCHK(cg.gmax.reserve((size_t)c.K * sizeof(double)));
        if (c.cheb.nlev > 0 && (cg.shape == SH_VOIGT || cg.shape == SH_LORENTZ) &&
            (rc = gas_interp_build(ctx, cg.itp, c.cheb, G.h_nu, g0, g1, nu, nnu, cg.cut, c.K, s, false)))
            return rc;
        HIPCHK(hipStreamSynchronize(s));   // J0, J1, win are locals
        maxL = std::max(maxL, (size_t)G.L);
    }
    if (c.cheb.nlev > 0) {
        const size_t fb = (size_t)c.cheb.nItot * CS_NC * cheb_kpad(K) * sizeof(double);
        if (c.chebF.bytes < fb) {
            HIPCHK(c.chebF.reserve(fb));
            HIPCHK(hipMemsetAsync(c.chebF.p, 0, c.chebF.bytes, s));   // padding states stay finite
        }
    }
    HIPCHK(c.hot.reserve(((size_t)K * maxL + 4) * sizeof(LineHot)));
    HIPCHK(c.cold.reserve((size_t)K * maxL * sizeof(LineCold)));
    HIPCHK(c.sigma.reserve((size_t)K * nnu * sizeof(double)));
    bool any_voigt = false;
    for (auto &cg : c.gas) any_voigt = any_voigt || cg.shape == SH_VOIGT;
    if (any_voigt) HIPCHK(c.sigma2.reserve((size_t)K * nnu * sizeof(double)));
    if (ngas > 0) HIPCHK(c.ranges.reserve((size_t)K * nnu * sizeof(int2) + (size_t)2 * K * ((nnu + 63) / 64) * sizeof(int)));   // + per-(tile, state) flags
    HIPCHK(c.tau.reserve((size_t)nl * nnu * sizeof(double)));   // the caller's output, or k_flux_chunk's scratch between its two sweeps
    if (!c.ticket.p) {
        HIPCHK(c.ticket.reserve((size_t)(1 + 512 / CS_FLUX_GROUP) * sizeof(unsigned)));
        HIPCHK(hipMemsetAsync(c.ticket.p, 0, c.ticket.bytes, s));
    }
    if (c.want_M) {
        HIPCHK(c.Mup.reserve((size_t)np * nnu * sizeof(double)));
        HIPCHK(c.Mdn.reserve((size_t)np * nnu * sizeof(double)));
    }
    HIPCHK(c.partial.reserve(((size_t)std::max<int64_t>(c.rtg.nblk, (nnu + 63) / 64) + 512 / CS_FLUX_GROUP) * 2 * np * sizeof(double)));   // (one block per tile at most) + k_flux's group sums
    HIPCHK(c.F.reserve((size_t)2 * np * sizeof(double)));
    HIPCHK(hipStreamSynchronize(s));
    c.ready = true;  // state upload below needs the sizes
    c.tab.clear();
    c.cia.clear();
    c.accel.slot = -1;
    if ((rc = cs_column_update_state(ctx, T_nodes, mu_nodes, T_levels, conc, nullptr))) { c.ready = false; return rc; }
    return CS_OK;
}

int cs_column_update_state(cs_ctx *ctx, const double *T_nodes, const double *mu_nodes, const double *T_levels,
                           const double *conc, const double *conc_tab)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "cs_column_setup has not been called");
    Column &c = ctx->col;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int K = c.K, nl = c.nl, nlob = c.nlob;
    std::vector<double> Tk(K), muk(K);
    Tk[0] = T_nodes[0];
    muk[0] = mu_nodes[0];
    for (int i = 0; i < nl; i++)
        for (int n = 1; n < nlob; n++) {
            Tk[i * (nlob - 1) + n] = T_nodes[n + (size_t)nlob * i];
            muk[i * (nlob - 1) + n] = mu_nodes[n + (size_t)nlob * i];
        }
    int rc;
    c.h_Tk = Tk;
    for (int gi = 0; gi < c.ngas; gi++)
        if ((rc = check_gas_states(ctx->gas[c.ugas[gi].slot], K, Tk.data()))) return rc;
    if ((rc = upload(c.Tk, Tk.data(), K, s)) || (rc = upload(c.muk, muk.data(), K, s)) ||
        (rc = upload(c.Tlev, T_levels, c.np, s)))
        return rc;
    for (auto &cg : c.gas) {   // per group: concentration and partial pressure of every member [nmem][K], Lorentz-width bound over the members
        const int nm = (int)cg.mem.size();
        std::vector<double> cc((size_t)nm * K), pp((size_t)nm * K), gb(K, 0.0);
        for (int m = 0; m < nm; m++) {
            const int gi = cg.mem[m];
            for (int k = 0; k < K; k++) {
                const double v = conc[gi + (size_t)c.ngas * k];
                if (!(v >= 0 && v <= 1))
                    return fail(CS_EINVAL, "gas molar concentrations must be in [0,1], not %g", v);
                cc[(size_t)m * K + k] = v;
                pp[(size_t)m * K + k] = v * c.h_Pk[k];  // Pp = C*P, gases.jl:126
            }
            const std::vector<double> gm = gamma_bound(ctx->gas[c.ugas[gi].slot], K, Tk.data(), c.h_Pk.data(), pp.data() + (size_t)m * K);
            for (int k = 0; k < K; k++) gb[k] = std::max(gb[k], gm[k]);
        }
        std::vector<double> lrt, qr;
        state_tables(*cg.tab, K, Tk.data(), lrt, qr);
        if ((rc = upload(cg.conc, cc.data(), cc.size(), s)) || (rc = upload(cg.Pp, pp.data(), pp.size(), s)) ||
            (rc = upload(cg.gmax, gb.data(), K, s)) || (rc = upload(cg.lrt, lrt.data(), lrt.size(), s)) || (rc = upload(cg.qref, qr.data(), qr.size(), s)))
            return rc;
        HIPCHK(hipStreamSynchronize(s));   // (locals)
    }
    HIPCHK(hipStreamSynchronize(s));
    if (!c.tab.empty()) {
        if (!conc_tab) return fail(CS_EINVAL, "the resident column has opacity tables: conc_tab is required");
        if ((rc = upload_tables(ctx, conc_tab))) return rc;
    }
    return CS_OK;
}

int cs_column_batch(cs_ctx *ctx, int B, const double *T_nodes, const double *mu_nodes, const double *T_levels,
                    const double *conc, const double *conc_tab, const double *cia_P1, const double *cia_P2, double *Fup, double *Fdn)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "cs_column_setup has not been called");
    Column &c = ctx->col;
    if (B < 1 || B > 65535) return fail(CS_EINVAL, "batch size must be in [1, 65535]");
    {
        const int rc0 = column_current(ctx);
        if (rc0) return rc0;
    }
    if (!c.tab.empty() && !conc_tab) return fail(CS_EINVAL, "the resident column has opacity tables: conc_tab is required");
    if (!c.cia.empty() && (!cia_P1 || !cia_P2)) return fail(CS_EINVAL, "the resident column has CIA pairs: cia_P1 and cia_P2 are required");
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int K = c.K, nl = c.nl, nlob = c.nlob, np = c.np;
    const int64_t BK = (int64_t)B * K;
    const size_t nn = (size_t)nlob * nl;
    std::vector<double> Tk(BK), muk(BK), Pk(BK);
    for (int b = 0; b < B; b++) {
        const double *Tn = T_nodes + b * nn, *mn = mu_nodes + b * nn;
        double *t = Tk.data() + (size_t)b * K, *m = muk.data() + (size_t)b * K;
        t[0] = Tn[0];
        m[0] = mn[0];
        for (int i = 0; i < nl; i++)
            for (int n = 1; n < nlob; n++) {
                t[i * (nlob - 1) + n] = Tn[n + (size_t)nlob * i];
                m[i * (nlob - 1) + n] = mn[n + (size_t)nlob * i];
            }
        std::copy(c.h_Pk.begin(), c.h_Pk.end(), Pk.begin() + (size_t)b * K);
    }
    int rc;
    for (int gi = 0; gi < c.ngas; gi++)
        if ((rc = check_gas_states(ctx->gas[c.ugas[gi].slot], (int)BK, Tk.data()))) return rc;
    const double *extra = c.has_extra ? c.extra.as<double>() : nullptr;
    if (extra) return fail(CS_EINVAL, "host-evaluated sigma(nu,T,P) terms are not supported in batch mode");
    const bool shared_sigma = c.accel.slot >= 0;   // AcceleratedAbsorber: cross-sections do not depend on the thermal state (absorbers.jl:203)
    DevBuf dTk, dPk, dmuk, dTlev, dsig, dtau, dpart, dF, dranges, dconc, dPp, dgb, dzones, dizones, dF2, dsep, dedge, hot, cold, dlrt, dqref;
    if ((rc = upload(dTk, Tk.data(), BK, s)) || (rc = upload(dPk, Pk.data(), BK, s)) || (rc = upload(dmuk, muk.data(), BK, s)) ||
        (rc = upload(dTlev, T_levels, (size_t)B * np, s)))
        return rc;
    const RtGeom bg = rt_geometry(c.nnu, np, B, c.nstream, ctx->tune[5] != 0);
    if (!shared_sigma) HIPCHK(dsig.reserve((size_t)BK * c.nnu * sizeof(double)));
    HIPCHK(dpart.reserve((size_t)B * bg.nblk * 2 * np * sizeof(double)));
    HIPCHK(dF.reserve((size_t)B * 2 * np * sizeof(double)));
    double *sig = shared_sigma ? c.sigma.as<double>() : dsig.as<double>();
    if (shared_sigma) {
        int e = 0;
        if ((rc = sigma_impl(ctx, s, nullptr, e))) return rc;
    } else if (c.ngas == 0) {
        const int64_t tot = BK * c.nnu;
        CS_LAUNCH(k_fill, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, tot, c.sigma_gray, (const double *)nullptr, sig);
    }
    size_t maxL = 0;
    for (auto &g : c.gas) maxL = std::max(maxL, (size_t)g.tab->L);
    const size_t per_state = maxL * (sizeof(LineHot) + sizeof(LineCold) + (ctx->mixed ? sizeof(LineF32) : 0)) + (size_t)c.nnu * sizeof(int2);
    const int kc = (int)std::max<size_t>(1, std::min<size_t>({(size_t)BK, ((size_t)8 << 30) / std::max<size_t>(per_state, 1), (size_t)65535}));
    if (c.ngas > 0) {
        HIPCHK(hot.reserve(((size_t)kc * maxL + 4) * sizeof(LineHot)));
        HIPCHK(cold.reserve((size_t)kc * maxL * sizeof(LineCold)));
        HIPCHK(dranges.reserve((size_t)kc * c.nnu * sizeof(int2) + (size_t)2 * kc * ((c.nnu + 63) / 64) * sizeof(int)));
        if (ctx->mixed) HIPCHK(ctx->hot32.reserve(((size_t)kc * maxL + 4) * sizeof(LineF32)));
    }
    for (auto &cg : c.gas)
        if (cg.shape == SH_PHCO2) { ph_set_grid(ctx, ctx->ph, c.h_nu.data(), c.nnu, c.grid_id); break; }
    for (size_t qi = 0; qi < c.gas.size(); qi++) {
        ColGas &cg = c.gas[qi];
        const GasTable &G = *cg.tab;
        const int nm = (int)cg.mem.size();
        std::vector<double> cc((size_t)nm * BK), pp((size_t)nm * BK), gb(BK, 0.0);
        for (int m = 0; m < nm; m++) {
            const int gi = cg.mem[m];
            double *cm = cc.data() + (size_t)m * BK, *pm = pp.data() + (size_t)m * BK;
            for (int b = 0; b < B; b++)
                for (int k = 0; k < K; k++) {
                    const double v = conc[(size_t)b * c.ngas * K + gi + (size_t)c.ngas * k];
                    if (!(v >= 0 && v <= 1)) return fail(CS_EINVAL, "gas molar concentrations must be in [0,1], not %g", v);
                    cm[(size_t)b * K + k] = v;
                    pm[(size_t)b * K + k] = v * c.h_Pk[k];
                }
            const std::vector<double> gm = gamma_bound(ctx->gas[c.ugas[gi].slot], (int)BK, Tk.data(), Pk.data(), pm);
            for (int64_t k = 0; k < BK; k++) gb[k] = std::max(gb[k], gm[k]);
        }
        const size_t nt64 = (size_t)((c.nnu + 63) / 64);
        std::vector<double> lrt, qr;
        state_tables(G, (int)BK, Tk.data(), lrt, qr);
        if ((rc = upload(dconc, cc.data(), cc.size(), s)) || (rc = upload(dPp, pp.data(), pp.size(), s)) || (rc = upload(dgb, gb.data(), BK, s)) ||
            (rc = upload(dlrt, lrt.data(), lrt.size(), s)) || (rc = upload(dqref, qr.data(), qr.size(), s)))
            return rc;
        HIPCHK(dzones.reserve((size_t)kc * nt64 * sizeof(Zone)));
        Interp itp;
        if (cg.itp.nlev > 0) {
            HIPCHK(dizones.reserve((size_t)kc * c.cheb.nItot * sizeof(IZone)));
            const size_t fb = (size_t)c.cheb.nItot * CS_NC * cheb_kpad(kc) * sizeof(double);
            if (dF2.bytes < fb) {
                HIPCHK(dF2.reserve(fb));
                HIPCHK(hipMemsetAsync(dF2.p, 0, dF2.bytes, s));
            }
            itp = interp_view(c.cheb, cg.itp, kc, dizones.as<IZone>());
            itp.F = dF2.as<double>();
            HIPCHK(dsep.reserve((size_t)((kc + 15) / 16) * c.cheb.nItot * sizeof(SepZone)));   // (the column's own buffer is sized for K states)
            HIPCHK(dedge.reserve((size_t)((kc + 15) / 16) * nt64 * sizeof(EdgeZone)));
            itp.sep = ctx->matrix_nodes ? dsep.as<SepZone>() : nullptr;
            itp.edge = ctx->matrix_nodes ? dedge.as<EdgeZone>() : nullptr;
            itp.sep_always = ctx->matrix_nodes == 2;
            interp_settings(ctx, itp);
            itp.core = ctx->matrix_core != 0;
        }
        for (int64_t k0 = 0; k0 < BK; k0 += kc) {
            const int kn = (int)std::min<int64_t>(kc, BK - k0);
            launch_gas(s, cg.shape, G, cg.jlo, cg.jhi, kn, dTk.as<double>() + k0, dPk.as<double>() + k0, dPp.as<double>() + k0,
                       dconc.as<double>() + k0, (int)BK, dlrt.as<double>() + k0, dqref.as<double>() + (size_t)k0 * G.niso, hot.as<LineHot>(), cold.as<LineCold>(), c.nu.as<double>(), c.nnu, c.ntile,
                       cg.J0.as<int32_t>(), cg.J1.as<int32_t>(), cg.win.as<WaveWin>(), cg.xtiles, dzones.as<Zone>(), dranges.as<int2>(),
                       dgb.as<double>() + k0, cg.cut, c.sigma_gray, nullptr, sig + (size_t)k0 * c.nnu, qi > 0, nullptr,
                       (ctx->mixed && cg.shape == SH_VOIGT) ? ctx->hot32.as<LineF32>() : nullptr, ctx->far_s, itp, nullptr, &ctx->ph);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(s));   // cc/pp/gb are locals; the device buffers are reused by the next group
    }
    // baked gases of the column at all B*K states: the Gas functor fC(T,P)*exp(Phi(T, ln P)) (gases.jl:85,278) -- what RCM holds
    // inside its AcceleratedAbsorber (radiative_convective.jl:6-103)
    if (!shared_sigma && !c.tab.empty()) {
        const int nt = (int)c.tab.size();
        std::vector<double> W, ct(BK);
        DevBuf dW, dct;
        for (int t = 0; t < nt; t++) {
            TableDev &tb = ctx->tab[c.tab[t].slot];
            if ((rc = table_weights(tb, (int)BK, Tk.data(), Pk.data(), W))) return rc;
            for (int b = 0; b < B; b++)
                for (int k = 0; k < K; k++) {
                    const double v = conc_tab[(size_t)b * nt * K + t + (size_t)nt * k];
                    if (!(v >= 0 && v <= 1)) return fail(CS_EINVAL, "gas molar concentrations must be in [0,1], not %g", v);
                    ct[(size_t)b * K + k] = v;
                }
            if ((rc = upload(dW, W.data(), W.size(), s)) || (rc = upload(dct, ct.data(), BK, s))) return rc;
            const int M = tb.nT * tb.nP;
            if ((rc = launch_table_eval(s, tb.Z.as<double>(), M, c.nnu, dW.as<double>(), (int)BK, dct.as<double>(), sig))) return rc;
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(s));   // W/ct host and device buffers are reused by the next table
        }
    }
    if (!shared_sigma && !c.cia.empty()) {
        const int nc = (int)c.cia.size();
        DevBuf dst, d1, d2, da;
        std::vector<double> p1(BK), p2(BK);
        for (int t = 0; t < nc; t++) {
            ColCia &ci = c.cia[t];
            for (int b = 0; b < B; b++)
                for (int k = 0; k < K; k++) {
                    p1[(size_t)b * K + k] = cia_P1[(size_t)b * nc * K + t + (size_t)nc * k];
                    p2[(size_t)b * K + k] = cia_P2[(size_t)b * nc * K + t + (size_t)nc * k];
                }
            if ((rc = upload_cia_states(ctx, ci.slot, ci.flags, (int)BK, Tk.data(), Pk.data(), p1.data(), p2.data(), 1, 0, dst, d1, d2, da))) return rc;
            CS_LAUNCH(k_cia, dim3((unsigned)c.ntile), dim3(256), 0, s, ci.nband, ci.bands.as<CiaBand>(), dst.as<CiaState>(),
                               c.nu.as<double>(), c.nnu, (int)BK, d1.as<double>(), d2.as<double>(), da.as<double>(), sig);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(s));
        }
    }
    launch_rt(c.nstream, bg, B, s, c.rt, c.nu.as<double>(),
                    c.wts.as<double>(), c.nnu, sig, dmuk.as<double>(), c.P.as<double>(), dTlev.as<double>(),
                    c.has_S ? c.S_toa.as<double>() : nullptr, c.has_alb ? c.albedo.as<double>() : nullptr, (double *)nullptr, nullptr,
                    nullptr, dpart.as<double>(), shared_sigma ? 0 : (size_t)K * c.nnu);   // batches return band fluxes only: no tau stored
    CS_LAUNCH(k_freduce, dim3(2 * np, B), dim3(256), 0, s, dpart.as<double>(), bg.nblk, 2 * np, dF.as<double>());
    HIPCHK(hipGetLastError());
    std::vector<double> F((size_t)B * 2 * np);
    HIPCHK(hipMemcpyAsync(F.data(), dF.p, F.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int b = 0; b < B; b++) {
        std::copy(F.begin() + (size_t)b * 2 * np, F.begin() + (size_t)b * 2 * np + np, Fup + (size_t)b * np);
        std::copy(F.begin() + (size_t)b * 2 * np + np, F.begin() + (size_t)(b + 1) * 2 * np, Fdn + (size_t)b * np);
    }
    return CS_OK;
}

// the cross-section stage of one evaluation: sigma[K][nnu] of all absorbers of the resident column at its node states
// (Sigma(A, i, T, P) of absorbers.jl:95 for every i and node).  ev: see run_impl; e counts the events recorded.
// near_plane_live: NULL = the cross-sections themselves are the result (the near-line plane is folded into sigma before returning);
// else the caller (run_impl) hands both planes to k_rt and is told here whether the second one is in use this step
// fuse != NULL: the interpolated wings are not carried to the grid and the CIA pairs are not added here -- *fuse says what the flux
// kernel still has to add (k_flux; near_plane_live must be given too: the near-line plane is not folded in either)
static int sigma_impl(cs_ctx *ctx, hipStream_t s, hipEvent_t *ev, int &e, bool *near_plane_live, FluxFuse *fuse)
{
    Column &c = ctx->col;
    const int K = c.K;
    {
        const int rc0 = column_current(ctx);
        if (rc0) return rc0;
    }
    double *sig = c.sigma.as<double>();
    const double *extra = c.has_extra ? c.extra.as<double>() : nullptr;
    if (c.accel.slot >= 0) {   // AcceleratedAbsorber: exp of the ln P-interpolated ln sigma (absorbers.jl:203)
        AccelDev &ad = ctx->accel[c.accel.slot];
        CS_LAUNCH(k_accel_eval, dim3((unsigned)c.ntile, K), dim3(256), 0, s, ad.L.as<double>(), c.nnu, K, c.accel.cell.as<int32_t>(),
                           c.accel.x.as<double>(), c.accel.xa.as<double>(), c.accel.xb.as<double>(), c.sigma_gray, extra, sig);
    } else if (c.gas.empty()) {
        const int64_t tot = (int64_t)K * c.nnu;
        CS_LAUNCH(k_fill, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, tot, c.sigma_gray, extra, sig);
    }
    if (ctx->mixed) {
        size_t maxL = 0;
        for (auto &g : c.gas) maxL = std::max(maxL, (size_t)g.tab->L);
        HIPCHK(ctx->hot32.reserve(((size_t)K * maxL + 4) * sizeof(LineF32)));   // no-op once sized (not capturable the first time)
    }
    ChebApply apply;
    apply.ngas = 0;
    for (auto &cg : c.gas)
        if (cg.shape == SH_PHCO2) { ph_set_grid(ctx, ctx->ph, c.h_nu.data(), c.nnu, c.grid_id); break; }
    int n_itp = 0;
    for (auto &cg : c.gas) n_itp += cg.itp.nlev > 0 ? 1 : 0;
    // cs_set_tuning key 2: node sums on a side stream -- 1: where the grid is short (fewer than 16384 (tile, state) waves), 2: always
    // key 7: the near-line kernels on a second side stream, into a plane of their own (1, default; 0: after k_voigt_edge_mx, into sigma)
    Fork fk;
    memset(&fk, 0, sizeof fk);
    fk.s2 = ctx->stream2; fk.ev_fork = ctx->ev_fork; fk.ev_join = ctx->ev_join;
    fk.s3 = ctx->stream3; fk.ev_fork3 = ctx->ev_fork3; fk.ev_join3 = ctx->ev_join3; fk.ev_far3 = ctx->ev_far3;
    fk.use_nodes = !ev && (ctx->tune[2] == 2 || (ctx->tune[2] == 1 && (c.nnu + 63) / 64 * (int64_t)K < 16384));
    // (1 = where it was measured to pay: 1/8 shards of C3 -3..-6 %, C3 -1 %; not on tiny columns -- C2 +19 %: the join costs more than
    //  the kernels -- nor on very long sparse ones -- C5 +2 %; 2 = always)
    const int64_t tswaves = (c.nnu + 63) / 64 * (int64_t)K;
    fk.use_near = !ev && c.sigma2.p != nullptr && (ctx->tune[7] == 2 || (ctx->tune[7] == 1 && tswaves >= 8192 && tswaves <= 300000));
    fk.sigma2 = fk.use_near ? c.sigma2.as<double>() : nullptr;
    const bool use_fork = fk.use_nodes || fk.use_near;
    for (int gi = 0; gi < (int)c.gas.size(); gi++) {
        ColGas &cg = c.gas[gi];
        const GasTable &G = *cg.tab;
        Interp itp = cg.itp.nlev > 0 ? interp_view(c.cheb, cg.itp, K) : Interp();
        itp.F = c.chebF.as<double>();
        if (!ctx->matrix_nodes) itp.sep = nullptr, itp.edge = nullptr;
        itp.sep_always = ctx->matrix_nodes == 2;
        interp_settings(ctx, itp);
        itp.core = ctx->matrix_core != 0;
        itp.fuse_apply = ctx->tune[0] != 0 && n_itp == 1;
        launch_gas(s, cg.shape, G, cg.jlo, cg.jhi, K, c.Tk.as<double>(), c.Pk.as<double>(), cg.Pp.as<double>(), cg.conc.as<double>(), K,
                   cg.lrt.as<double>(), cg.qref.as<double>(), c.hot.as<LineHot>(), c.cold.as<LineCold>(), c.nu.as<double>(), c.nnu, c.ntile, cg.J0.as<int32_t>(),
                   cg.J1.as<int32_t>(), cg.win.as<WaveWin>(), cg.xtiles, cg.zones.as<Zone>(), c.ranges.as<int2>(), cg.gmax.as<double>(), cg.cut, c.sigma_gray, extra, sig, gi > 0,
                   ev ? ev + e : nullptr,
                   (ctx->mixed && cg.shape == SH_VOIGT) ? ctx->hot32.as<LineF32>() : nullptr, ctx->far_s, itp, &apply, &ctx->ph,
                   use_fork ? &fk : nullptr);
        if (ev) { e += 6; HIPCHK(hipEventRecord(ev[e++], s)); }
    }
    if (fuse) { fuse->apply = 0; fuse->ncia = 0; fuse->Kpad = cheb_kpad(K); }
    // short grids with the node sums on their side stream: that stream has slack (it ends well before the per-point kernels), so the levels
    // are folded into the smallest one THERE (k_cheb_cascade, exact to rounding) and the flux kernel carries one level to the grid instead
    // of three -- one memory latency in its first phase instead of three (13 -> 6 us on a 1/8 shard of the bench column)
    bool cascaded_aside = false;
    ChebApply carried;   // what a cascade on the side stream leaves to carry to the grid
    const int nl_itp = apply.ngas == 1 ? apply.nlev - apply.l0[0] : 0;
    const bool short_aside = fuse && c.rtg.streams;
    // long grids: where the cascade is in use anyway (four levels and up) it runs on the node-sum stream as well, beside the per-point
    // kernels, instead of between them and the flux kernel (BASELINE configs[4]: four launches, 0.32 ms of the main stream)
    // ... and with the node sums on a side stream the cascade is off the step's critical path whatever the number of levels: that stream
    // ends long before the other two at full size (the bench column: 1.34 against 1.60 and 1.86 ms into the step), so folding three levels
    // into one there makes the carry to the grid a third of its matrix products and of its reads of C (round 5: 1.880 -> 1.872 ms)
    const bool casc_on = nl_itp >= 2 && ctx->tune[12] != 2 && (ctx->tune[12] == 1 || cascade_pays(nl_itp) || fk.pending);
    if (apply.ngas == 1 && fk.pending && !(ctx->tune[15] & 256) && nl_itp >= 2 && (short_aside || casc_on)) {
        const double *Rc[CS_MAX_LEVEL];
        for (int l = 0; l < CS_MAX_LEVEL; l++) Rc[l] = c.cheb.Rc[l].as<double>();
        launch_apply_cascade(fk.s2, apply, Rc, c.cheb.itv, c.cheb.nI, 1, cheb_kpad(K), c.nnu, K, 0.0, nullptr, sig, 1, &carried);
        (void)hipEventRecord(fk.ev_join, fk.s2);   // (the main stream has not waited yet: it will wait for this later record)
        cascaded_aside = true;
    }
    fork_join(&fk, s, true, false);   // the node sums; the near-line kernels may run on beside what follows (none of it touches their plane)
    // interpolated far wings of all gases: sigma += sum_level C (sum_gas F)  (one pass over C and sigma)
    if (apply.ngas > 0 && cascaded_aside) {
        if (fuse) { fuse->A = carried; fuse->apply = 1; }
        else launch_apply(s, carried, cheb_kpad(K), c.nnu, K, 0.0, nullptr, sig, 1);
    } else if (apply.ngas > 0) {
        const double *Rc[CS_MAX_LEVEL];
        for (int l = 0; l < CS_MAX_LEVEL; l++) Rc[l] = c.cheb.Rc[l].as<double>();
        launch_apply_cascade(s, apply, Rc, c.cheb.itv, c.cheb.nI, ctx->tune[12], cheb_kpad(K), c.nnu, K, 0.0, nullptr, sig, 1,
                             fuse ? &fuse->A : nullptr);
        if (fuse) fuse->apply = 1;
    }
    for (auto &t : c.tab) {  // baked gases: sigma += fC * exp(Phi(T, ln P))
        TableDev &tb = ctx->tab[t.slot];
        int rc2;
        if ((rc2 = launch_table_eval(s, tb.Z.as<double>(), tb.nT * tb.nP, c.nnu, t.W.as<double>(), K, t.conc.as<double>(), sig))) return rc2;
    }
    for (auto &cc : c.cia) {  // CIA pairs
        if (fuse) {   // the temperature half of the interpolation per (band, state, sample); k_flux does the rest per point
            CiaPairDev &pd = fuse->cia[fuse->ncia++];
            pd.nband = cc.nband; pd.bands = cc.bands.as<CiaBand>(); pd.st = cc.st.as<CiaState>(); pd.tab = cc.tab.as<double>();
            pd.toff = cc.toff.as<int64_t>(); pd.rho1 = cc.rho1.as<double>(); pd.rho2 = cc.rho2.as<double>(); pd.rhoa = cc.rhoa.as<double>();
            pd.tband = cc.tband.as<int32_t>(); pd.cell = cc.cell.as<int32_t>(); pd.fx = cc.fx.as<double>();
            pd.nslot = std::min(std::max(cc.max_overlap, 1), (int)CS_CIA_ACT);
            pd.fac = cc.fac.as<double>();
            int maxnb = 0;
            for (auto &b : ctx->cia[cc.slot].bands) maxnb = std::max(maxnb, (int)b.nu.size());
            CS_LAUNCH(k_cia_tab, dim3((unsigned)((maxnb + 255) / 256), (unsigned)K, (unsigned)cc.nband), dim3(256), 0, s, pd, K);
            continue;
        }
        CS_LAUNCH(k_cia, dim3((unsigned)c.ntile), dim3(256), 0, s, cc.nband, cc.bands.as<CiaBand>(), cc.st.as<CiaState>(),
                           c.nu.as<double>(), c.nnu, K, cc.rho1.as<double>(), cc.rho2.as<double>(), cc.rhoa.as<double>(), sig);
    }
    fork_join(&fk, s);
    c.near_live = false;
    c.sigma_partial = fuse != nullptr;
    if (near_plane_live) *near_plane_live = fk.live;
    else if (fk.live) {
        const int64_t tot = (int64_t)K * c.nnu;
        CS_LAUNCH(k_fold, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, tot, sig, c.sigma2.as<double>());
    }
    HIPCHK(hipGetLastError());
    return CS_OK;
}

// a table re-uploaded into (or cleared from) a slot of the resident column leaves its windows stale: refuse to run on them
static int column_current(cs_ctx *ctx)
{
    Column &c = ctx->col;
    for (auto &ug : c.ugas)
        if (!ctx->gas[ug.slot].present || ctx->gas[ug.slot].generation != ug.generation) {
            c.ready = false;
            return fail(CS_ESTATE, "gas slot %d was re-uploaded or cleared after cs_column_setup", ug.slot);
        }
    // the same for the opacity tables and the accelerated absorber the column names by slot: its weights [nT * nP][K] / knot cells were
    // formed for the (T, P) grid / knots the slot held at cs_column_set_tables / cs_column_set_accel
    for (auto &ct : c.tab)
        if (!ctx->tab[ct.slot].present || ctx->tab[ct.slot].generation != ct.generation)
            return fail(CS_ESTATE, "opacity-table slot %d was baked, uploaded or cleared after cs_column_set_tables: call it again", ct.slot);
    if (c.accel.slot >= 0 && (!ctx->accel[c.accel.slot].present || ctx->accel[c.accel.slot].generation != c.accel.generation))
        return fail(CS_ESTATE, "accelerated-absorber slot %d got other knots (or was cleared) after cs_column_set_accel: call it again", c.accel.slot);
    return CS_OK;
}

// enqueue one evaluation; when ev != NULL an event is recorded between the kernel classes (ev must hold 7 * (launch groups) + 4)
static int run_impl(cs_ctx *ctx, hipStream_t s, hipEvent_t *ev)
{
    Column &c = ctx->col;
    double *sig = c.sigma.as<double>();
    int e = 0, rc;
    g_nlaunch = 0;
    g_near_launches = 0;
    g_line_kernel = 0;
    c.last_stream = s;
    if (ev) HIPCHK(hipEventRecord(ev[e++], s));
    bool near_live = false;
    size_t fsh = 0;
    int fblk = 0, fthr = 0;
    const int form = flux_form(ctx, c, &fsh, &fblk, &fthr);
    FluxFuse fuse;
    memset(&fuse, 0, sizeof fuse);
    if ((rc = sigma_impl(ctx, s, ev, e, &near_live, form ? &fuse : nullptr))) return rc;
    c.near_live = near_live;
    c.sigma_partial = form != 0;
    c.flux_form_last = form;
    if (ev) HIPCHK(hipEventRecord(ev[e++], s));
    const double *dS = c.has_S ? c.S_toa.as<double>() : nullptr, *dA = c.has_alb ? c.albedo.as<double>() : nullptr;
    double *dMu = c.want_M ? c.Mup.as<double>() : nullptr, *dMd = c.want_M ? c.Mdn.as<double>() : nullptr;
    bool reduced = false;
    if (form) {
        fuse.sigma2 = near_live ? c.sigma2.as<double>() : nullptr;
        fuse.F = c.flux_out();
        // up to 512 blocks the last ones to finish add the block partials (two stages of 16 and <= 32 terms); longer grids keep k_freduce
        fuse.ticket = (fblk <= 512 && !(ctx->tune[15] & 4) && ctx->gfx950) ? c.ticket.as<unsigned>() : nullptr;   // (| 4: k_freduce always, for A/B)
        fuse.gpartial = c.partial.as<double>() + (size_t)std::max<int64_t>(c.rtg.nblk, (c.nnu + 63) / 64) * 2 * c.np;
        reduced = fuse.ticket != nullptr;
        if ((ctx->tune[15] & 128) && c.fluxdbg.reserve((8 + 2 * (size_t)fblk + 32) * sizeof(unsigned long long)) == hipSuccess) fuse.dbg = c.fluxdbg.as<unsigned long long>();
        // the chunked form always writes the layer optical depths (its upward sweep reads them back): into the caller's plane or scratch
        double *dtau = (c.want_tau || form == 2) ? c.tau.as<double>() : nullptr;   // (forms 1 and 3 keep the optical depths in LDS)
#define CS_FLUX_CASE(N) case N: launch_flux_ns<N>(form, fsh, fblk, fthr, s, c.rt, c.nu.as<double>(), c.wts.as<double>(), c.nnu, sig, c.muk.as<double>(), \
                                                  c.P.as<double>(), c.Tlev.as<double>(), dS, dA, dtau, dMu, dMd, c.partial.as<double>(), fuse, (ctx->tune[15] & 8) == 0, (ctx->tune[15] & 2048) != 0); break;
        switch (c.nstream) {
            CS_FLUX_CASE(1) CS_FLUX_CASE(2) CS_FLUX_CASE(3) CS_FLUX_CASE(4) CS_FLUX_CASE(5) CS_FLUX_CASE(6) CS_FLUX_CASE(7) CS_FLUX_CASE(8)
            CS_FLUX_CASE(9) CS_FLUX_CASE(10) CS_FLUX_CASE(11) CS_FLUX_CASE(12) CS_FLUX_CASE(13) CS_FLUX_CASE(14) CS_FLUX_CASE(15) CS_FLUX_CASE(16)
        }
#undef CS_FLUX_CASE
    } else {
        launch_rt(c.nstream, c.rtg, 1, s, c.rt, c.nu.as<double>(), c.wts.as<double>(),
                  c.nnu, sig, c.muk.as<double>(), c.P.as<double>(), c.Tlev.as<double>(), dS, dA, c.want_tau ? c.tau.as<double>() : nullptr,
                  dMu, dMd, c.partial.as<double>(), 0, near_live ? c.sigma2.as<double>() : nullptr);
    }
    if (ev) HIPCHK(hipEventRecord(ev[e++], s));
    if (!reduced)
        CS_LAUNCH(k_freduce, dim3(2 * c.np), dim3(256), 0, s, c.partial.as<double>(), form ? fblk : c.rtg.nblk, 2 * c.np, c.flux_out());
    if (ev) HIPCHK(hipEventRecord(ev[e++], s));
    c.launches = g_nlaunch;
    c.near_launches_last = g_near_launches;
    c.line_kernel_last = g_line_kernel;
    HIPCHK(hipGetLastError());
    return CS_OK;
}

int cs_column_sigma_run(cs_ctx *ctx, void *stream)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "cs_column_setup has not been called");
    HIPCHK(hipSetDevice(ctx->device));
    int e = 0;
    return sigma_impl(ctx, stream ? (hipStream_t)stream : ctx->stream, nullptr, e);
}

int cs_column_run(cs_ctx *ctx, void *stream)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "cs_column_setup has not been called");
    HIPCHK(hipSetDevice(ctx->device));   // (a process may drive several contexts on several devices: cs_fluxes_discretized_multi)
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    Column &c = ctx->col;
    if (!ctx->tune[4]) return run_impl(ctx, s, nullptr);
    if (c.graph_exec) {
        {   // (a replayed graph carries the kernel arguments of the slots' old contents: gas tables, opacity tables, knots)
            const int rc0 = column_current(ctx);
            if (rc0) { drop_graph(c); return rc0; }
        }
        HIPCHK(hipGraphLaunch(c.graph_exec, s));
        // the replayed kernels wrote the near-line plane again and ran on THIS stream: what cs_column_sigma_fetch / cs_column_fetch /
        // cs_column_info read must say so, as after an eager run
        c.near_live = c.graph_near_live;
        c.sigma_partial = c.graph_sigma_partial;
        c.launches = c.graph_launches;
        c.last_stream = s;
        return CS_OK;
    }
    if (c.runs_since_change++ == 0) return run_impl(ctx, s, nullptr);   // first run after a change: eager (workspaces may still grow)
    HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
    const int rc = run_impl(ctx, s, nullptr);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    if (rc || e != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        c.runs_since_change = 0;
        if (rc) return rc;
        return run_impl(ctx, s, nullptr);   // (capture refused: run eagerly)
    }
    c.graph = g;
    c.graph_near_live = c.near_live;   // (set by the captured run_impl)
    c.graph_sigma_partial = c.sigma_partial;
    c.graph_launches = c.launches;
    if (hipGraphInstantiate(&c.graph_exec, g, nullptr, nullptr, 0) != hipSuccess) { c.graph_exec = nullptr; drop_graph(c); return run_impl(ctx, s, nullptr); }
    HIPCHK(hipGraphLaunch(c.graph_exec, s));
    return CS_OK;
}

int cs_column_profile(cs_ctx *ctx, void *stream, int reps, double *ms)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "cs_column_setup has not been called");
    if (reps < 1 || !ms) return fail(CS_EINVAL, "bad arguments");
    Column &c = ctx->col;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIPCHK(hipSetDevice(ctx->device));
    const int ngrp = (int)c.gas.size();
    const int nev = 7 * ngrp + 4;
    std::vector<hipEvent_t> ev(nev);
    for (auto &e : ev) HIPCHK(hipEventCreate(&e));
    for (int i = 0; i < 10; i++) ms[i] = 0.0;
    int rc = CS_OK;
    for (int r = 0; r < reps && rc == CS_OK; r++) {
        rc = run_impl(ctx, s, ev.data());
        if (rc) break;
        if (hipStreamSynchronize(s) != hipSuccess) { rc = fail(CS_EHIP, "hipStreamSynchronize failed"); break; }
        float t;
        const int slot[7] = {0, 1, 7, 3, 9, 8, 4};   // per gas: K1 + zones, nodes, nodes on the matrix cores, far, sub-tile cores, far on the matrix cores, near
        for (int gi = 0; gi < ngrp; gi++)
            for (int q = 0; q < 7; q++) { (void)hipEventElapsedTime(&t, ev[7 * gi + q], ev[7 * gi + q + 1]); ms[slot[q]] += t; }
        const int b = 7 * ngrp;
        (void)hipEventElapsedTime(&t, ev[b], ev[b + 1]); ms[2] += t;       // apply (+ baked tables, CIA)
        (void)hipEventElapsedTime(&t, ev[b + 1], ev[b + 2]); ms[5] += t;   // rt
        (void)hipEventElapsedTime(&t, ev[b + 2], ev[b + 3]); ms[6] += t;   // reduce
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    for (int i = 0; i < 10; i++) ms[i] /= reps;
    return rc;
}

int cs_column_sync(cs_ctx *ctx)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    return CS_OK;
}

int cs_column_flux_ptr(cs_ctx *ctx, double **dF)
{
    if (!ctx || !ctx->col.ready || !dF) return fail(CS_ESTATE, "no resident column");
    *dF = ctx->col.flux_out();
    return CS_OK;
}

int cs_column_flux_to(cs_ctx *ctx, double *dst_device, void *stream)
{
    if (!ctx || !ctx->col.ready || !dst_device) return fail(CS_ESTATE, "no resident column");
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    if (dst_device != ctx->col.flux_out())   // (the kernels write there already: cs_column_set_flux_dst)
        HIPCHK(hipMemcpyAsync(dst_device, ctx->col.flux_out(), (size_t)2 * ctx->col.np * sizeof(double), hipMemcpyDeviceToDevice, s));
    return CS_OK;
}

int cs_column_set_flux_dst(cs_ctx *ctx, double *dst_device)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "no resident column");
    if (ctx->col.F_dst != dst_device) drop_graph(ctx->col);   // (a captured step carries the old pointer)
    ctx->col.F_dst = dst_device;
    return CS_OK;
}

static int fetch_transposed(cs_ctx *ctx, const double *dsrc, int R, int64_t Cn, double *hdst)
{
    Column &c = ctx->col;
    hipStream_t s = ctx->stream;
    HIPCHK(c.stage.reserve((size_t)R * Cn * sizeof(double)));
    dim3 grid((unsigned)((Cn + 31) / 32), (unsigned)((R + 31) / 32));
    CS_LAUNCH(k_transpose, grid, dim3(256), 0, s, dsrc, R, Cn, c.stage.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(hdst, c.stage.p, (size_t)R * Cn * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return CS_OK;
}

int cs_column_fetch(cs_ctx *ctx, int64_t nnu, int np, double *tau, double *Mup, double *Mdn, double *Fup, double *Fdn)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "no resident column");
    Column &c = ctx->col;
    if (nnu != c.nnu || np != c.np)   // the caller's buffers were sized for another column: refuse rather than overrun them
        return fail(CS_ESTATE, "resident column is %lld wavenumbers x %d levels, caller expects %lld x %d (another column was set up on this context)",
                    (long long)c.nnu, c.np, (long long)nnu, np);
    HIPCHK(hipSetDevice(ctx->device));
    // wait for the column's own work only (its run's stream; the side stream has been joined into it): other contexts on the same
    // device -- cs_fluxes_discretized_multi with several ranges per card -- keep computing while this one copies back
    if (c.last_stream && c.last_stream != ctx->stream) HIPCHK(hipStreamSynchronize(c.last_stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int rc;
    if (tau && !c.want_tau) return fail(CS_ESTATE, "column was set up without want_tau");
    if (tau && (rc = fetch_transposed(ctx, c.tau.as<double>(), c.nl, c.nnu, tau))) return rc;
    if ((Mup || Mdn) && !c.want_M) return fail(CS_ESTATE, "column was set up without want_M");
    if (Mup && (rc = fetch_transposed(ctx, c.Mup.as<double>(), c.np, c.nnu, Mup))) return rc;
    if (Mdn && (rc = fetch_transposed(ctx, c.Mdn.as<double>(), c.np, c.nnu, Mdn))) return rc;
    if (Fup) HIPCHK(hipMemcpyAsync(Fup, c.flux_out(), c.np * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (Fdn) HIPCHK(hipMemcpyAsync(Fdn, c.flux_out() + c.np, c.np * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

int cs_column_sigma_fetch(cs_ctx *ctx, int64_t nnu, int K, double *sigma)
{
    if (!ctx || !ctx->col.ready || !sigma) return fail(CS_ESTATE, "no resident column");
    Column &c = ctx->col;
    if (nnu != c.nnu || K != c.K)
        return fail(CS_ESTATE, "resident column is %lld wavenumbers x %d node states, caller expects %lld x %d", (long long)c.nnu, c.K,
                    (long long)nnu, K);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    if (c.sigma_partial) {   // the run finished the cross-sections on chip (k_flux): evaluate them once more, all the way into HBM
        int e = 0;
        const int rc = sigma_impl(ctx, ctx->stream, nullptr, e);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(ctx->stream));
        c.sigma_partial = false;
        c.near_live = false;
    }
    if (c.near_live) {   // the run handed k_rt two planes: the total is their sum (folded in once)
        const int64_t tot = (int64_t)c.K * c.nnu;
        CS_LAUNCH(k_fold, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, tot, c.sigma.as<double>(), c.sigma2.as<double>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));
        c.near_live = false;
    }
    HIPCHK(hipMemcpy(sigma, c.sigma.p, (size_t)c.K * c.nnu * sizeof(double), hipMemcpyDeviceToHost));
    return CS_OK;
}

int cs_column_counts(cs_ctx *ctx, int64_t *pair_evals, int64_t *lines_in_range)
{
    if (!ctx || !ctx->col.ready) return fail(CS_ESTATE, "no resident column");
    Column &c = ctx->col;
    int64_t p = 0, l = 0;
    for (auto &g : c.ugas) {
        if (g.pairs_per_state < 0) g.pairs_per_state = count_pairs(ctx->gas[g.slot].h_nu, c.h_nu.data(), c.nnu, g.cut);
        p += g.pairs_per_state * c.K;
        l += g.lines_in_range;
    }
    if (pair_evals) *pair_evals = p;
    if (lines_in_range) *lines_in_range = l;
    return CS_OK;
}

int cs_column_info(cs_ctx *ctx, int64_t *out)
{
    if (!ctx || !ctx->col.ready || !out) return fail(CS_ESTATE, "no resident column");
    const Column &c = ctx->col;
    for (int i = 0; i < 8; i++) out[i] = 0;
    out[0] = (int64_t)c.gas.size();
    out[1] = c.launches;
    for (auto &g : c.gas) { out[2] += g.tab->L; out[4] = std::max<int64_t>(out[4], (int64_t)g.mem.size()); }
    out[3] = c.merge;
    out[5] = c.flux_form_last;
    out[6] = c.near_launches_last;
    out[7] = c.line_kernel_last;
    return CS_OK;
}

int cs_interp_plan(int64_t nnu, const double *nu, double dnu_cut, int *interval_sizes)
{
    if (!nu || nnu < 1 || !interval_sizes) return fail(CS_EINVAL, "bad arguments");
    return choose_levels(nu, nnu, dnu_cut, interval_sizes);   // (default size range; a context's cs_set_interp_plan may narrow it)
}

int cs_phco2_plan(int64_t nnu, const double *nu, double dnu_cut, int cap, int *interval_sizes, int *nodes, int *regions)
{
    if (!nu || nnu < 1 || !interval_sizes || !nodes || !regions || cap < 1) return fail(CS_EINVAL, "bad arguments");
    cs_ctx defaults;          // (host data only: the default interpolation settings)
    PhScratch ph;
    ph_set_grid(&defaults, ph, nu, nnu, 1);
    const int nv = ph_plan(&ph, nnu, dnu_cut);
    for (int v = 0; v < std::min(nv, cap); v++) {
        interval_sizes[v] = ph.grid.lv.itv[ph.grid.vl.rl[v]];
        nodes[v] = ph.grid.vl.nc[v];
        regions[v] = ph.grid.vl.rmask[v];
    }
    return nv;
}

int cs_column_work(cs_ctx *ctx, int64_t *out)
{
    if (!ctx || !ctx->col.ready || !out) return fail(CS_ESTATE, "no resident column");
    Column &c = ctx->col;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipDeviceSynchronize());
    const int K = c.K;
    const int nt64 = (int)((c.nnu + 63) / 64);
    int64_t direct = 0, nodes = 0, sepn = 0, edgen = 0, mx3 = 0, subn = 0, ncore = 0, mx8 = 0, mx3n = 0;   // subn: (lane, line) evaluations of k_voigt_sub
    //   // sepn, edgen: (node | point, line, state) triples summed on the matrix cores; mx3: those with 3 terms
    // flops of the two matrix-core kernels: issued = every matrix instruction's 2048; useful = 2 x terms per (column, line, state) with
    // the column inside the cut-off, outside the core radius, and the state a real one (a group's tail rows are padding)
    double fl_edge_useful = 0.0, fl_edge_issued = 0.0, fl_nodes_useful = 0.0, fl_nodes_issued = 0.0, fl_apply = 0.0;
    int64_t rec_edge = 0, rec_nodes = 0;   // (state, line) records the two matrix-core kernels REQUEST: lines of every piece x 16 states (neighbouring tiles and
                                           // intervals ask for the same record again: the unique ones are K x lines in range)
    int64_t near0 = 0, near1 = 0;
    int64_t body[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // per-point lines by body: 2-term, 2-term+cut-off, 3-term, 3-term+cut-off, 4-term+cut-off,
                                                     // near-zone pass; node lines: 2-, 3-, 4-term
    auto seg = [](int lo, int hi, int p0, int p1) { return (int64_t)std::max(0, std::min(hi, p1) - std::max(lo, p0)); };
    for (auto &g : c.gas) {
        if (g.shape != SH_VOIGT && g.shape != SH_LORENTZ) continue;
        std::vector<WaveWin> win(nt64);
        std::vector<Zone> zn((size_t)K * nt64);
        HIPCHK(hipMemcpy(win.data(), g.win.p, win.size() * sizeof(WaveWin), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(zn.data(), g.zones.p, zn.size() * sizeof(Zone), hipMemcpyDeviceToHost));
        std::vector<IZone> iz;
        const int nlev = g.itp.nlev, nItot = c.cheb.nItot;
        // nodes a piece of interval q is summed on in k_cheb_nodes_mx: 16 or 32 for the far pieces (p = 0, 3) of the intervals that are
        // not shared by the four waves of a block, 64 otherwise (the carry to the 64 nodes, 2 x 64 x n per state, is not counted)
        int nsplit_w = 0;
        if (nlev > 0) {
            const int q0w = c.cheb.ioff[g.itp.l0], nqw = nItot - q0w;
            for (int l = g.itp.l0; l < std::min(nlev, g.itp.l0 + (ctx->tune[6] > 0 ? ctx->tune[6] : 1)); l++) nsplit_w += c.cheb.nI[l];
            if (!mx_big(nqw, K, 2048) || nsplit_w > nqw) nsplit_w = nqw;
            nsplit_w += q0w;   // intervals below this index are split
        }
        auto far_nodes = [&](int q, int p) {
            // (since round 5 the shared items take their far pieces on fewer nodes as well; key 17: not where every item is shared)
            if (ctx->tune[11] || (p != 0 && p != 3) || (ctx->tune[17] && nsplit_w >= nItot)) return (int)CS_NC;
            int l = 0;
            while (l + 1 < nlev && q >= c.cheb.ioff[l + 1]) l++;
            return g.itp.nfar[l] > 0 ? g.itp.nfar[l] : (int)CS_NC;
        };
        if (nlev > 0) {
            iz.resize((size_t)K * nItot);
            HIPCHK(hipMemcpy(iz.data(), g.itp.iz.p, iz.size() * sizeof(IZone), hipMemcpyDeviceToHost));
            const int q0 = c.cheb.ioff[g.itp.l0];
            const bool use_sep = sep_in_use(ctx->matrix_nodes != 0, ctx->matrix_nodes == 2, nItot - q0, K, g.shape != SH_VOIGT, ctx->mixed != 0, ctx->tune[1] != 0);
            std::vector<SepZone> sz;
            if (use_sep) {
                sz.resize((size_t)((K + 15) / 16) * nItot);
                HIPCHK(hipMemcpy(sz.data(), g.itp.sep.p, sz.size() * sizeof(SepZone), hipMemcpyDeviceToHost));
            }
            for (int k = 0; k < K; k++)
                for (int q = q0; q < nItot; q++) {
                    const IZone &z = iz[(size_t)k * nItot + q];
                    nodes += (int64_t)CS_NC * ((z.P0 - z.E0) + (z.Z0 - z.P1) + (z.P2 - z.Z1) + (z.E1 - z.P3));
                    // the same segments k_cheb_nodes runs: [E0,P0) U [P1,Z0) left, [Z1,P2) U [P3,E1) right, each minus the piece
                    // the matrix-core kernel takes, cut at Q and M
                    int sa[4] = {z.P0, z.Z0, z.Z1, z.P3}, sb[4] = {z.P0, z.Z0, z.Z1, z.P3};
                    if (use_sep) {
                        const SepZone &s4 = sz[(size_t)(k >> 4) * nItot + q];
                        for (int p = 0; p < 4; p++) {   // of the group's piece, this state's part: the lines beyond its own series radius (k_cheb_nodes_mx's mask)
                            const int pa = p < 2 ? s4.a[p] : std::max(s4.a[p], z.S1), pb = p < 2 ? std::min(s4.b[p], z.S0) : s4.b[p];
                            if (pb > pa) {
                                sa[p] = pa; sb[p] = pb; sepn += (int64_t)CS_NC * (pb - pa);
                                const int n3 = p < 2 ? std::max(0, std::min(s4.m[p], pb) - pa) : std::max(0, pb - std::max(s4.m[p], pa));
                                mx3 += (int64_t)CS_NC * n3;
                                mx3n += (int64_t)CS_NC * n3;
                                fl_nodes_useful += 2.0 * far_nodes(q, p) * (3.0 * n3 + 4.0 * ((pb - pa) - n3));
                            }
                        }
                    }
                    const int lo8[8] = {z.E0, sb[0], z.P1, sb[1], sb[3], z.P3, sb[2], z.Z1}, hi8[8] = {sa[0], z.P0, sa[1], z.Z0, z.E1, sa[3], z.P2, sa[2]};
                    for (int w8 = 0; w8 < 8; w8++) {
                        const int p0 = lo8[w8], p1 = hi8[w8];
                        if (w8 < 4) {
                            body[6] += seg(z.E0, z.Q0, p0, p1); body[7] += seg(z.Q0, z.M0, p0, p1); body[8] += seg(z.M0, z.Z0, p0, p1);
                        } else {
                            body[6] += seg(z.Q1, z.E1, p0, p1); body[7] += seg(z.M1, z.Q1, p0, p1); body[8] += seg(z.Z1, z.M1, p0, p1);
                        }
                    }
                }
        }
        if (nlev > 0) {
            const int q0 = c.cheb.ioff[g.itp.l0];
            if (sep_in_use(ctx->matrix_nodes != 0, ctx->matrix_nodes == 2, nItot - q0, K, g.shape != SH_VOIGT, ctx->mixed != 0, ctx->tune[1] != 0)) {
                std::vector<SepZone> sz((size_t)((K + 15) / 16) * nItot);
                HIPCHK(hipMemcpy(sz.data(), g.itp.sep.p, sz.size() * sizeof(SepZone), hipMemcpyDeviceToHost));
                for (int gq = 0; gq < (K + 15) / 16; gq++) {
                    const int ns = std::min(16, K - 16 * gq);
                    for (int q = q0; q < nItot; q++) {
                        const SepZone &z = sz[(size_t)gq * nItot + q];
                        for (int p = 0; p < 4; p++) {
                            if (z.b[p] <= z.a[p]) continue;
                            const int n3 = p < 2 ? z.m[p] - z.a[p] : z.b[p] - z.m[p], n4 = (z.b[p] - z.a[p]) - n3;
                            (void)ns;   // (useful flops: per state, above -- a state takes part only beyond its own series radius)
                            rec_nodes += 16 * (int64_t)(n3 + n4);
                            fl_nodes_issued += 2.0 * far_nodes(q, p) * 16 * (3.0 * ((n3 + 3) / 4 * 4) + 4.0 * ((n4 + 3) / 4 * 4));
                        }
                    }
                }
            }
        }
        int ishift = 0;
        if (nlev > 0) for (int r = c.cheb.itv[nlev - 1] / 64; r > 1; r >>= 1) ishift++;
        const bool use_edge = nlev > 0 && edge_in_use(ctx->matrix_nodes != 0, ctx->matrix_nodes == 2, nt64, K, g.shape != SH_VOIGT, ctx->mixed != 0,
                                                      std::max<int64_t>(g.jhi - g.jlo, 0), ctx->tune[1] != 0);
        std::vector<EdgeZone> ez;
        if (use_edge) {
            ez.resize((size_t)((K + 15) / 16) * nt64);
            HIPCHK(hipMemcpy(ez.data(), g.itp.edge.p, ez.size() * sizeof(EdgeZone), hipMemcpyDeviceToHost));
        }
        if (use_edge) {
            const double *nl = g.tab->h_nu.data(), *vv = c.h_nu.data();
            for (int gq = 0; gq < (K + 15) / 16; gq++) {
                const int ns = std::min(16, K - 16 * gq);
                for (int t = 0; t < nt64; t++) {
                    const WaveWin w = win[t];
                    const EdgeZone e = ez[(size_t)gq * nt64 + t];
                    const double *v0 = vv + (size_t)t * 64, *v1 = vv + std::min<int64_t>((int64_t)t * 64 + 64, c.nnu);
                    auto piece = [&](int ja, int jb, int nt, int mask) {   // mask 0: every point counts; 1: |dnu| <= cut; 2: also |dnu| >= R
                        if (jb <= ja) return;
                        rec_edge += 16 * (int64_t)(jb - ja);
                        fl_edge_issued += 2.0 * nt * 64.0 * 16.0 * ((jb - ja + 3) / 4 * 4);
                        double cols = 0.0;
                        for (int j = ja; j < jb; j++) {
                            if (mask == 0) { cols += (double)(v1 - v0); continue; }
                            int n = (int)(std::upper_bound(v0, v1, nl[j] + g.cut) - std::lower_bound(v0, v1, nl[j] - g.cut));
                            if (mask == 2) n -= (int)(std::lower_bound(v0, v1, nl[j] + e.R) - std::upper_bound(v0, v1, nl[j] - e.R));   // (|dnu| < R: k_voigt_sub's)
                            cols += std::max(n, 0);
                        }
                        fl_edge_useful += 2.0 * nt * cols * ns;
                    };
                    // the cut-off edges, cut where the next sub-tile comes into reach (k_voigt_edge_mx's phases: one wave per (tile, group) only)
                    const bool phased = !ctx->tune[14] && mx_big(nt64, K, 1024);
                    bool tile_carry = false;
                    auto end_piece = [&](int ja, int jb, int nt, bool left) {
                        if (jb <= ja) return;
                        // the lines inside the cut-off of every point of the tile: on 16 nodes of the tile (one sub-tile per step) + the carry
                        // to the points, 16 matrix instructions per (tile, group) that has any
                        const bool tnodes_on = c.cheb.tile_nodes_ok && !ctx->tune[23];
                        if (!phased || jb - ja < 48) { piece(ja, jb, nt, 1); return; }
                        const double issued0 = fl_edge_issued;
                        piece(ja, jb, nt, 1);              // (for the useful flops)
                        fl_edge_issued = issued0;
                        const double tolc = 1e-9 * (std::fabs(v0[0]) + g.cut + 1.0);
                        int cutp[5];
                        if (left) {
                            if (tnodes_on) {
                                const int j3 = (int)(std::lower_bound(nl + ja, nl + jb, *(v1 - 1) - g.cut + tolc) - nl);
                                if (jb - j3 >= 16) {
                                    fl_edge_issued += 2.0 * nt * 16.0 * 16.0 * ((jb - j3 + 3) / 4 * 4);
                                    tile_carry = true;
                                    jb = j3;
                                }
                            }
                            cutp[0] = ja; cutp[4] = jb;
                            for (int q = 0; q < 3; q++) {
                                const double *col = vv + std::min<int64_t>((int64_t)t * 64 + 16 * (q + 1), c.nnu - 1);
                                cutp[q + 1] = (int)(std::lower_bound(nl + ja, nl + jb, *col - g.cut - tolc) - nl);
                            }
                            for (int q = 1; q < 5; q++) cutp[q] = std::max(cutp[q], cutp[q - 1]);
                            for (int q = 0; q < 4; q++) fl_edge_issued += 2.0 * nt * 16.0 * (q + 1) * 16.0 * ((cutp[q + 1] - cutp[q] + 3) / 4 * 4);
                        } else {
                            if (tnodes_on) {
                                const int u3 = (int)(std::upper_bound(nl + ja, nl + jb, v0[0] + g.cut - tolc) - nl);
                                if (u3 - ja >= 16) {
                                    fl_edge_issued += 2.0 * nt * 16.0 * 16.0 * ((u3 - ja + 3) / 4 * 4);
                                    tile_carry = true;
                                    ja = u3;
                                }
                            }
                            cutp[0] = ja; cutp[4] = jb;
                            for (int q = 0; q < 3; q++) {
                                const double *col = vv + std::min<int64_t>((int64_t)t * 64 + 16 * q + 15, c.nnu - 1);
                                cutp[q + 1] = (int)(std::upper_bound(nl + ja, nl + jb, *col + g.cut + tolc) - nl);
                            }
                            for (int q = 1; q < 5; q++) cutp[q] = std::max(cutp[q], cutp[q - 1]);
                            for (int q = 0; q < 4; q++) fl_edge_issued += 2.0 * nt * 16.0 * (4 - q) * 16.0 * ((cutp[q + 1] - cutp[q] + 3) / 4 * 4);
                        }
                    };
                    tile_carry = false;
                    end_piece(w.W0, e.eL, (e.far3 & 1) ? 3 : 4, true);
                    end_piece(e.eR, w.W1, (e.far3 & 2) ? 3 : 4, false);
                    if (tile_carry) fl_edge_issued += 16.0 * 2048.0;
                    if (e.mL1 > e.mL0) { piece(e.mL0, e.mL3, 3, 1); piece(e.mL3, e.mL1, 4, 1); }
                    if (e.mR1 > e.mR0) { piece(e.mR3, e.mR1, 3, 1); piece(e.mR0, e.mR3, 4, 1); }
                    if (e.cR > e.cL) piece(e.cL, e.cR, (e.far3 & 4) ? 8 : 4, 2);
                }
            }
        }
        if (nlev > 0)   // the contraction that carries the node sums to the grid (k_cheb_apply_mfma, or fused into k_voigt_edge_mx)
            fl_apply += 2.0 * CS_NC * 64.0 * 16.0 * ((K + 15) / 16) * (double)nt64 * (nlev - g.itp.l0);   // (per group; with the cascade: cs_column_info)
        for (int k = 0; k < K; k++)
            for (int t = 0; t < nt64; t++) {
                WaveWin w = win[t];
                const Zone &z = zn[(size_t)k * nt64 + t];
                const int W0 = w.W0, W1 = w.W1;
                int pm[4] = {0, 0, 0, 0};   // [pL0, pL1), [pR0, pR1): the pieces between interpolated sets and near zone on the matrix cores
                int cc0 = 0, cc1 = 0;       // [cc0, cc1): the core that k_voigt_sub and the second mask of k_voigt_edge_mx share
                if (use_edge) {   // what k_voigt_edge_mx takes
                    const EdgeZone e = ez[(size_t)(k >> 4) * nt64 + t];
                    edgen += 64 * (int64_t)((e.eL - W0) + (W1 - e.eR));
                    mx3 += 64 * (int64_t)(((e.far3 & 1) ? e.eL - W0 : 0) + ((e.far3 & 2) ? W1 - e.eR : 0));
                    w.W0 = e.eL; w.W1 = e.eR;
                    if (e.mL1 > e.mL0) { pm[0] = e.mL0; pm[1] = e.mL1; mx3 += 64 * (int64_t)(e.mL3 - e.mL0); }
                    if (e.mR1 > e.mR0) { pm[2] = e.mR0; pm[3] = e.mR1; mx3 += 64 * (int64_t)(e.mR1 - e.mR3); }
                    edgen += 64 * (int64_t)((pm[1] - pm[0]) + (pm[3] - pm[2]));
                    if (e.cR > e.cL) {   // the core: every pair visits the matrix cores (masked inside R), k_voigt_sub the lines within R of each sub-tile
                        cc0 = e.cL; cc1 = e.cR;
                        ncore++;
                        edgen += 64 * (int64_t)(e.cR - e.cL);
                        if (e.far3 & 4) mx8 += 64 * (int64_t)(e.cR - e.cL);
                        const double *nl = g.tab->h_nu.data();
                        for (int q4 = 0; q4 < 64 / CS_SUBW; q4++) {   // (k_voigt_sub<CS_SUBW>)
                            const double v0 = c.h_nu[(size_t)t * 64 + CS_SUBW * q4] - e.R, v1 = c.h_nu[(size_t)t * 64 + CS_SUBW * q4 + CS_SUBW - 1] + e.R;
                            const int ja = (int)(std::lower_bound(nl + e.cL, nl + e.cR, v0) - nl);
                            const int jb = (int)(std::upper_bound(nl + ja, nl + e.cR, v1) - nl);
                            subn += CS_SUBW * (int64_t)(jb - ja);
                        }
                    }
                }
                int64_t n = (w.W1 - w.W0) - (pm[1] - pm[0]) - (pm[3] - pm[2]) - (cc1 - cc0);
                int sa0 = z.M0, sa1 = z.M0, sb0 = z.M1, sb1 = z.M1;
                if (nlev > 0) {   // same clamps as k_voigt_far
                    const IZone &zi = iz[(size_t)k * nItot + c.cheb.ioff[nlev - 1] + (t >> ishift)];
                    sa0 = std::min(std::max(zi.E0, W0), z.N0); sa1 = std::min(std::max(zi.Z0, sa0), z.N0);
                    sb0 = std::max(std::min(zi.Z1, W1), z.N1); sb1 = std::max(std::min(zi.E1, W1), sb0);
                    n -= (sa1 - sa0) + (sb1 - sb0);
                }
                direct += 64 * n;
                // the same segments k_voigt_far runs inside its three clip windows
                const int a = std::min(std::max(w.E0, W0), z.Q0), a1 = std::min(std::max(w.E0, z.Q0), z.M0);
                const int b1 = std::max(std::min(w.E1, z.Q1), z.M1), bq = std::max(std::min(w.E1, W1), z.Q1);
                const int pL0 = pm[1] > pm[0] ? pm[0] : sa1, pL1 = pm[1] > pm[0] ? pm[1] : sa1;
                const int pR0 = pm[3] > pm[2] ? pm[2] : sb0, pR1 = pm[3] > pm[2] ? pm[3] : sb0;
                const int cM0 = cc1 > cc0 ? cc0 : pR0, cM1 = cc1 > cc0 ? cc1 : pR0;
                const int cl[6] = {w.W0, sa1, pL1, cM1, pR1, sb1}, ch[6] = {sa0, pL0, cM0, pR0, sb0, w.W1};
                for (int cw = 0; cw < 6; cw++) {
                    const int p0 = cl[cw], p1 = ch[cw];
                    if (p0 >= p1) continue;
                    body[1] += seg(W0, a, p0, p1) + seg(bq, W1, p0, p1);
                    body[0] += seg(a, z.Q0, p0, p1) + seg(z.Q1, bq, p0, p1);
                    body[3] += seg(z.Q0, a1, p0, p1) + seg(b1, z.Q1, p0, p1);
                    body[2] += seg(a1, z.M0, p0, p1) + seg(z.M1, b1, p0, p1);
                    body[4] += seg(z.M0, z.N0, p0, p1) + seg(z.N1, z.M1, p0, p1);
                    body[5] += seg(z.N0, z.N1, p0, p1);
                }
            }
    }
    // near-line pairs by tier (k_voigt_near<0>: 100 <= s < 1e3, <1>: s < 100), counted from the records of the launch group whose
    // per-(state, line) records are still in HBM -- the last Voigt group of the column (the only one when its gases are merged)
    if (!c.gas.empty() && c.gas.back().shape == SH_VOIGT && c.gas.back().jhi > c.gas.back().jlo) {
        const ColGas &g = c.gas.back();
        const int64_t L = g.tab->L, nj = g.jhi - g.jlo;
        std::vector<LineHot> rec((size_t)nj);
        const double *vv = c.h_nu.data();
        const int64_t nnu = c.nnu;
        for (int k = 0; k < K; k++) {
            HIPCHK(hipMemcpy(rec.data(), c.hot.as<LineHot>() + (size_t)k * L + g.jlo, rec.size() * sizeof(LineHot), hipMemcpyDeviceToHost));
            for (int64_t j = 0; j < nj; j++) {
                const LineHot &h = rec[j];
                auto within = [&](double smax) -> int64_t {   // points with x^2 + y^2 < smax and |dnu| <= cut
                    if (!(h.p2 < smax)) return 0;
                    const double r = std::min(std::sqrt(smax - h.p2) / h.p1, g.cut);
                    return (std::lower_bound(vv, vv + nnu, h.nul + r) - std::upper_bound(vv, vv + nnu, h.nul - r));
                };
                const int64_t n1 = within(kMidS), n0 = within(kSerS);
                near1 += n1;
                near0 += n0 - n1;
            }
        }
    }
    out[20] = near0;
    out[21] = near1;
    out[22] = (int64_t)fl_edge_useful;
    out[23] = (int64_t)fl_edge_issued;
    out[24] = (int64_t)fl_nodes_useful;
    out[25] = (int64_t)fl_nodes_issued;
    out[26] = (int64_t)fl_apply;
    if (c.fluxdbg.p && (ctx->tune[15] & 128)) {   // k_flux_scan's phases in block 0, ns: cross-sections, depths + Planck, chunk pass, hand-over, second pass, band sum
        unsigned long long st[8] = {};
        (void)hipMemcpy(st, c.fluxdbg.p, sizeof st, hipMemcpyDeviceToHost);
        for (int q = 0; q < 4; q++) out[27 + q] = (int64_t)(st[q + 1] - st[q]) * 10;   // (100 MHz clock)
        out[31] = (int64_t)(st[7] - st[0]) * 10;   // block 0's first instruction to the last block's last word
        if (getenv("CS_FLUX_DBG") && c.flux_form_last == 3) {   // every block's start and end (k_flux_scan), relative to the earliest start
            const size_t nb = (size_t)(c.nnu + 63) / 64;
            if (c.fluxdbg.bytes >= (8 + 2 * nb + 32) * sizeof(unsigned long long)) {
                std::vector<unsigned long long> bs(2 * nb + 32);
                (void)hipMemcpy(bs.data(), (const char *)c.fluxdbg.p + 8 * sizeof(unsigned long long), (2 * nb + 32) * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                fprintf(stderr, "block 0 fine stamps [us from its start]: hand-over steps");
                for (int q = 0; q < 12; q++) fprintf(stderr, " %.1f", (double)(bs[2 * nb + q] - st[0]) * 0.01);
                fprintf(stderr, " | second pass down");
                for (int q = 0; q < 5; q++) fprintf(stderr, " %.1f", (double)(bs[2 * nb + 16 + q] - st[0]) * 0.01);
                fprintf(stderr, " up");
                for (int q = 4; q >= 0; q--) fprintf(stderr, " %.1f", (double)(bs[2 * nb + 24 + q] - st[0]) * 0.01);
                fprintf(stderr, "\n");
                unsigned long long t0 = ~0ull;
                for (size_t b = 0; b < nb; b++) t0 = std::min(t0, bs[2 * b]);
                std::vector<double> st0(nb), en0(nb), du(nb);
                for (size_t b = 0; b < nb; b++) { st0[b] = (double)(bs[2 * b] - t0) * 0.01; en0[b] = (double)(bs[2 * b + 1] - t0) * 0.01; du[b] = en0[b] - st0[b]; }
                auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (double)(v.size() - 1))]; };
                fprintf(stderr, "block 0 stamps [us from its start]:");
                for (int q = 1; q < 7; q++) fprintf(stderr, " %.1f", (double)(st[q] - st[0]) * 0.01);
                fprintf(stderr, "\n");
                fprintf(stderr, "flux blocks %zu [us]: start p50 %.1f p90 %.1f max %.1f | duration min %.1f p50 %.1f p90 %.1f max %.1f | end p50 %.1f p90 %.1f max %.1f | band fluxes stored %.1f\n",
                        nb, pct(st0, 0.5), pct(st0, 0.9), pct(st0, 1.0), pct(du, 0.0), pct(du, 0.5), pct(du, 0.9), pct(du, 1.0), pct(en0, 0.5), pct(en0, 0.9), pct(en0, 1.0),
                        (double)(st[7] - t0) * 0.01);
            }
        }
    } else {
        for (int q = 27; q < 32; q++) out[q] = 0;
    }
    out[32] = rec_edge * (int64_t)sizeof(LineHot);
    out[33] = rec_nodes * (int64_t)sizeof(LineHot);
    for (int q = 34; q < 40; q++) out[q] = 0;
    out[0] = direct;
    out[1] = nodes;
    out[2] = c.cheb.nlev;
    out[3] = c.cheb.nItot;
    for (int q = 0; q < 6; q++) out[4 + q] = 64 * body[q];
    for (int q = 6; q < 9; q++) out[4 + q] = (int64_t)CS_NC * body[q];
    out[13] = sepn;
    out[14] = edgen;
    out[15] = mx3;
    out[16] = subn;
    out[17] = ncore;
    out[18] = mx8;
    out[19] = mx3n;
    return CS_OK;
}

// true when the resident column was set up for exactly this grid, pressure levels, rule orders and gas line-up: only the
// thermal state and the per-call spectra differ, so a call can skip the window / interpolation-matrix / workspace setup
// (the members beyond line-by-line gases -- baked tables, CIA pairs, an accelerated absorber -- are named by their slots: ntab / ncia
//  = 0 and accel_slot = -1 for a column without them)
static bool column_matches(cs_ctx *ctx, int64_t nnu, const double *nu, int np, const double *P, double g, int nlobatto, int ngas,
                           const int *gas_slots, const int *shapes, const double *dnu_cuts, double sigma_gray, double theta_s,
                           int nstream, bool want_tau, bool want_M, int64_t g_nnu = 0, int64_t g_start = 0, double g_left = 0.0, double g_right = 0.0,
                           int ntab = 0, const int *table_slots = nullptr, int ncia = 0, const int *cia_slots = nullptr, int accel_slot = -1)
{
    const Column &c = ctx->col;
    const bool shard_ok = g_nnu > 0 ? (!c.default_wts && c.g_nnu == g_nnu && c.g_start == g_start && c.g_left == g_left && c.g_right == g_right)
                                    : c.default_wts;
    if (!c.ready || c.accel.slot != accel_slot || !shard_ok || c.interp != interp_key(ctx) || (int)c.tab.size() != ntab || (int)c.cia.size() != ncia) return false;
    for (int t = 0; t < ntab; t++)
        if (c.tab[t].slot != table_slots[t]) return false;
    for (int t = 0; t < ncia; t++)
        if (c.cia[t].slot != cia_slots[t]) return false;
    if (c.nnu != nnu || c.np != np || c.nlob != nlobatto || c.nstream != nstream || c.ngas != ngas) return false;
    if (c.g != g || c.sigma_gray != sigma_gray || c.theta_s != theta_s) return false;
    if (c.want_tau != want_tau || c.want_M != want_M) return false;
    if (c.merge != ctx->merge) return false;
    for (int gi = 0; gi < ngas; gi++) {
        const UserGas &cg = c.ugas[gi];
        if (cg.slot != gas_slots[gi] || cg.shape != (shapes ? shapes[gi] : CS_SHAPE_VOIGT) || cg.cut != (dnu_cuts ? dnu_cuts[gi] : 25.0)) return false;
        if (gas_slots[gi] < 0 || gas_slots[gi] >= CS_MAX_GAS || !ctx->gas[cg.slot].present || ctx->gas[cg.slot].generation != cg.generation) return false;
    }
    return memcmp(c.h_nu.data(), nu, (size_t)nnu * sizeof(double)) == 0 && memcmp(c.h_P.data(), P, (size_t)np * sizeof(double)) == 0;
}

int cs_fluxes_discretized(cs_ctx *ctx, int64_t nnu, const double *nu, int np, const double *P, double g, int nlobatto,
                          const double *T_nodes, const double *mu_nodes, const double *T_levels, int ngas,
                          const int *gas_slots, const int *shapes, const double *dnu_cuts, const double *conc,
                          double sigma_gray, const double *sigma_extra, const double *S_toa, const double *albedo,
                          double theta_s, int nstream, double *tau, double *Mup, double *Mdn, double *Fup, double *Fdn)
{
    return cs_fluxes_discretized_members(ctx, nnu, nu, np, P, g, nlobatto, T_nodes, mu_nodes, T_levels, ngas, gas_slots, shapes, dnu_cuts, conc,
                                         0, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, -1, sigma_gray, sigma_extra, S_toa, albedo,
                                         theta_s, nstream, tau, Mup, Mdn, Fup, Fdn);
}

int cs_fluxes_discretized_members(cs_ctx *ctx, int64_t nnu, const double *nu, int np, const double *P, double g, int nlobatto,
                                  const double *T_nodes, const double *mu_nodes, const double *T_levels, int ngas,
                                  const int *gas_slots, const int *shapes, const double *dnu_cuts, const double *conc,
                                  int ntab, const int *table_slots, const double *conc_tab,
                                  int ncia, const int *cia_slots, const int *cia_flags, const double *cia_P1, const double *cia_P2,
                                  int accel_slot, double sigma_gray, const double *sigma_extra, const double *S_toa, const double *albedo,
                                  double theta_s, int nstream, double *tau, double *Mup, double *Mdn, double *Fup, double *Fdn)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    if (!nu || !P || nnu < 1 || np < 2) return fail(CS_EINVAL, "bad grid arguments");
    if (ntab < 0 || ntab > CS_MAX_TABLE || (ntab > 0 && (!table_slots || !conc_tab))) return fail(CS_EINVAL, "bad opacity-table arguments");
    if (ncia < 0 || ncia > CS_MAX_CIA || (ncia > 0 && (!cia_slots || !cia_P1 || !cia_P2))) return fail(CS_EINVAL, "bad CIA arguments");
    if (accel_slot >= 0 && (ngas > 0 || ntab > 0 || ncia > 0))   // unifyabsorbers(::Tuple{AcceleratedAbsorber}), absorbers.jl:216
        return fail(CS_EINVAL, "a column over an accelerated absorber has no other absorbers");
    if (accel_slot < -1) accel_slot = -1;
    int rc;
    // radiate! is called once per time step / Jacobian column on an unchanged grid (radiative_convective.jl:109-171): keep the
    // column of the previous call resident and refresh only what a call can change -- node states, fS, fa, sigma_extra, and the
    // per-node inputs of the tables / CIA pairs / accelerated absorber it names
    if (column_matches(ctx, nnu, nu, np, P, g, nlobatto, ngas, gas_slots, shapes, dnu_cuts, sigma_gray, theta_s, nstream,
                       tau != nullptr, Mup || Mdn, 0, 0, 0.0, 0.0, ntab, table_slots, ncia, cia_slots, accel_slot)) {
        Column &c = ctx->col;
        HIPCHK(hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        const bool was[3] = {c.has_extra, c.has_S, c.has_alb};
        c.has_extra = sigma_extra != nullptr;
        c.has_S = S_toa != nullptr && std::any_of(S_toa, S_toa + nnu, [](double x) { return x != 0.0; });
        c.has_alb = albedo != nullptr && std::any_of(albedo, albedo + nnu, [](double x) { return x != 0.0; });
        if (was[0] != c.has_extra || was[1] != c.has_S || was[2] != c.has_alb) drop_graph(c);   // (other kernel arguments)
        if (c.has_extra && (rc = upload(c.extra, sigma_extra, (size_t)nnu * c.K, s))) return rc;
        if (c.has_S && (rc = upload(c.S_toa, S_toa, nnu, s))) return rc;
        if (c.has_alb && (rc = upload(c.albedo, albedo, nnu, s))) return rc;
        // (a table baked again into its slot since the last call may have another (T, P) grid: cs_column_set_tables re-checks it)
        if (ntab > 0 && (rc = cs_column_set_tables(ctx, ntab, table_slots, conc_tab))) return rc;
        if ((rc = cs_column_update_state(ctx, T_nodes, mu_nodes, T_levels, conc, ntab > 0 ? conc_tab : nullptr))) return rc;
    } else {
        rc = cs_column_setup(ctx, nnu, nu, nullptr, np, P, g, nlobatto, T_nodes, mu_nodes, T_levels, ngas, gas_slots,
                             shapes, dnu_cuts, conc, sigma_gray, sigma_extra, S_toa, albedo, theta_s, nstream,
                             tau != nullptr, (Mup || Mdn) ? 1 : 0);
        if (rc) return rc;
        if (ntab > 0 && (rc = cs_column_set_tables(ctx, ntab, table_slots, conc_tab))) return rc;
    }
    // per-node partial pressures of the CIA pairs and the knot cells of the accelerated absorber follow the node states: every call
    if (ncia > 0 && (rc = cs_column_set_cia(ctx, ncia, cia_slots, cia_flags, cia_P1, cia_P2))) return rc;
    if (accel_slot >= 0 && (rc = cs_column_set_accel(ctx, accel_slot))) return rc;
    if ((rc = cs_column_run(ctx, nullptr))) return rc;
    return cs_column_fetch(ctx, nnu, np, tau, Mup, Mdn, Fup, Fdn);
}

// ---- host-held reference objects handed over as they are ---------------------------------------------------------------------
int cs_table_upload(cs_ctx *ctx, int table_slot, int64_t nnu, const double *nu, int nT, const double *T, int nP, const double *P,
                    const double *lnsigma)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    if (table_slot < 0 || table_slot >= CS_MAX_TABLE) return fail(CS_EINVAL, "table slot %d out of range", table_slot);
    if (nT < 2 || nP < 2 || !T || !P || !lnsigma) return fail(CS_EINVAL, "need at least 2 x 2 grid points");
    int rc;
    if ((rc = check_ascending(nu, nnu))) return rc;
    for (int i = 1; i < nT; i++)
        if (!(T[i] > T[i - 1])) return fail(CS_EORDER, "table temperatures must be ascending");
    for (int j = 0; j < nP; j++)
        if (!(P[j] > 0) || (j > 0 && !(P[j] > P[j - 1]))) return fail(CS_EORDER, "table pressures must be positive and ascending");
    const size_t n = (size_t)nT * nP * nnu;
    for (size_t i = 0; i < n; i++)
        if (!std::isfinite(lnsigma[i])) return fail(CS_EINVAL, "ln sigma must be finite (OpacityTable stores ln(floatmin) for empty rows, gases.jl:76-80)");
    HIPCHK(hipSetDevice(ctx->device));
    TableDev &tb = ctx->tab[table_slot];
    tb.present = false;
    if ((rc = upload(tb.Z, lnsigma, n, ctx->stream))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    tb.nnu = nnu; tb.nT = nT; tb.nP = nP;
    tb.T.assign(T, T + nT);
    tb.lnP.resize(nP);
    for (int j = 0; j < nP; j++) tb.lnP[j] = std::log(P[j]);
    tb.nu.assign(nu, nu + nnu);
    tb.generation = next_generation();
    tb.present = true;
    return CS_OK;
}

int cs_accel_upload(cs_ctx *ctx, int accel_slot, int64_t nnu, const double *nu, int nk, const double *P_knots, const double *lnsigma)
{
    if (!ctx) return fail(CS_EINVAL, "ctx is NULL");
    if (accel_slot < 0 || accel_slot >= CS_MAX_ACCEL) return fail(CS_EINVAL, "accelerated-absorber slot %d out of range", accel_slot);
    if (nk < 2 || !P_knots || !lnsigma) return fail(CS_EINVAL, "need at least two pressure knots");
    int rc;
    if ((rc = check_ascending(nu, nnu))) return rc;
    for (int k = 0; k < nk; k++)
        if (!(P_knots[k] > 0) || (k > 0 && !(P_knots[k] > P_knots[k - 1]))) return fail(CS_EORDER, "knot pressures must be positive and strictly ascending");
    HIPCHK(hipSetDevice(ctx->device));
    AccelDev &ad = ctx->accel[accel_slot];
    const bool resident_here = ctx->col.ready && ctx->col.accel.slot == accel_slot;
    if (resident_here && (ad.nk != nk || ad.nnu != nnu)) ctx->col.ready = false;   // (its knot cells belong to the old knots)
    std::vector<double> lnP(nk);
    for (int k = 0; k < nk; k++) lnP[k] = std::log(P_knots[k]);
    // other knots (or another grid) than the slot held: a resident column's knot cells are stale (column_current refuses them until
    // cs_column_set_accel forms new ones); the same knots with new values -- what update! leaves -- keep the generation
    const bool same_knots = ad.present && ad.nk == nk && ad.nnu == nnu && ad.lnP == lnP && std::equal(ad.nu.begin(), ad.nu.end(), nu);
    ad.present = false;
    if ((rc = upload(ad.L, lnsigma, (size_t)nk * nnu, ctx->stream))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (!same_knots) ad.generation = next_generation();
    ad.nnu = nnu;
    ad.nk = nk;
    ad.nu.assign(nu, nu + nnu);
    ad.lnP = lnP;
    ad.present = true;
    return CS_OK;
}

int cs_accel_fetch(cs_ctx *ctx, int accel_slot, int64_t nnu, int nk, double *lnsigma)
{
    if (!ctx || accel_slot < 0 || accel_slot >= CS_MAX_ACCEL || !ctx->accel[accel_slot].present) return fail(CS_EINVAL, "accelerated-absorber slot is empty");
    const AccelDev &ad = ctx->accel[accel_slot];
    if (!lnsigma || nnu != ad.nnu || nk != ad.nk)
        return fail(CS_ESTATE, "the slot holds %lld wavenumbers x %d knots, caller expects %lld x %d", (long long)ad.nnu, ad.nk, (long long)nnu, nk);
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyAsync(lnsigma, ad.L.p, (size_t)nk * nnu * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

namespace {
struct MappedFile {
    const char *p = nullptr;
    size_t len = 0;
    int fd = -1;
    ~MappedFile()
    {
        if (p) munmap((void *)p, len);
        if (fd >= 0) close(fd);
    }
    int open_ro(const char *fn)
    {
        fd = ::open(fn, O_RDONLY);
        if (fd < 0) return fail(CS_EINVAL, "cannot open %s", fn);
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size == 0) return fail(CS_EINVAL, "cannot stat %s (or empty file)", fn);
        len = (size_t)st.st_size;
        void *m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return fail(CS_EINVAL, "cannot map %s", fn);
        p = (const char *)m;
        return CS_OK;
    }
};

// record length incl. line terminator: 160 characters + "\n" or "\r\n"
int par_layout(const MappedFile &f, size_t &stride, int64_t &n)
{
    const char *nl = (const char *)memchr(f.p, '\n', f.len);
    stride = nl ? (size_t)(nl - f.p) + 1 : f.len;
    const size_t body = (stride >= 2 && f.p[stride - 2] == '\r') ? stride - 2 : (nl ? stride - 1 : stride);
    if (body != 160) return fail(CS_EINVAL, "expected 160-character HITRAN records, found %zu", body);
    n = (int64_t)((f.len + (nl ? 0 : 1)) / stride);
    if ((size_t)n * stride < f.len && f.len - (size_t)n * stride >= 160) n++;  // last record without a terminator
    return CS_OK;
}

double field(const char *s, int a, int b)   // columns a..b (1-based, inclusive), like parse(Float64, line[a:b])
{
    char buf[32];
    const int w = b - a + 1;
    memcpy(buf, s + a - 1, w);
    buf[w] = 0;
    return strtod(buf, nullptr);
}
}  // namespace

int cs_par_count(const char *filename, int64_t *n)
{
    if (!filename || !n) return fail(CS_EINVAL, "bad arguments");
    MappedFile f;
    int rc = f.open_ro(filename);
    if (rc) return rc;
    size_t stride;
    return par_layout(f, stride, *n);
}

int cs_par_parse(const char *filename, int64_t n, int16_t *M, char *I, double *nu, double *S, double *A, double *gamma_a,
                 double *gamma_s, double *Epp, double *na, double *delta_a)
{
    if (!filename) return fail(CS_EINVAL, "bad arguments");
    MappedFile f;
    int rc = f.open_ro(filename);
    if (rc) return rc;
    size_t stride;
    int64_t cnt;
    if ((rc = par_layout(f, stride, cnt))) return rc;
    if (cnt != n) return fail(CS_EINVAL, "file holds %lld records, caller expects %lld", (long long)cnt, (long long)n);
    const int nth = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 32, n / 20000}));
    std::vector<std::thread> th;
    for (int t = 0; t < nth; t++)
        th.emplace_back([&, t]() {
            for (int64_t i = n * t / nth; i < n * (t + 1) / nth; i++) {
                const char *r = f.p + (size_t)i * stride;   // columns: par.jl:131-140
                M[i] = (int16_t)field(r, 1, 2);
                I[i] = r[2];
                nu[i] = field(r, 4, 15);
                S[i] = field(r, 16, 25);
                A[i] = field(r, 26, 35);
                gamma_a[i] = field(r, 36, 40);
                gamma_s[i] = field(r, 41, 45);
                Epp[i] = field(r, 46, 55);
                na[i] = field(r, 56, 59);
                delta_a[i] = field(r, 60, 67);
            }
        });
    for (auto &x : th) x.join();
    return CS_OK;
}

// ISOINDEX, par.jl:6-13: '1'..'9' -> 1..9, '0' -> 10, 'A'..'Z' -> 11..36; 0 = not an isotopologue character
static int iso_index(char c)
{
    if (c >= '1' && c <= '9') return c - '0';
    if (c == '0') return 10;
    if (c >= 'A' && c <= 'Z') return 11 + (c - 'A');
    return 0;
}

int cs_gas_upload_par(cs_ctx *ctx, int slot, const char *filename, double numin, double numax, double Scut, const int *iso_keep,
                      int n_iso_keep, int64_t maxlines, int M_expected, const double *mu_table, int niso, const int32_t *ncheb,
                      const double *cheb, int64_t *L_out)
{
    if (!ctx || !filename || !mu_table || !ncheb || !cheb || niso < 1) return fail(CS_EINVAL, "bad arguments");
    int64_t n = 0;
    int rc;
    if ((rc = cs_par_count(filename, &n))) return rc;
    std::vector<int16_t> M(n);
    std::vector<char> I(n);
    std::vector<double> nu(n), S(n), A(n), ga(n), gs(n), Epp(n), na(n), da(n);
    if ((rc = cs_par_parse(filename, n, M.data(), I.data(), nu.data(), S.data(), A.data(), ga.data(), gs.data(), Epp.data(), na.data(), da.data())))
        return rc;
    // par.jl:153-175: wavenumber range, intensity cut, isotopologue list
    std::vector<int64_t> keep;
    keep.reserve(n);
    for (int64_t i = 0; i < n; i++) {
        if (!(nu[i] >= numin && nu[i] <= numax && S[i] >= Scut)) continue;
        if (n_iso_keep > 0) {
            const int ii = iso_index(I[i]);
            bool ok = false;
            for (int q = 0; q < n_iso_keep; q++) ok = ok || iso_keep[q] == ii;
            if (!ok) continue;
        }
        keep.push_back(i);
    }
    if (keep.empty()) return fail(CS_EINVAL, "par information has been filtered to nothing!");   // par.jl:172
    // par.jl:177-186: the `maxlines` strongest lines (N is the unfiltered count there); order of equal intensities as the host
    // mirror produces it: ascending stable sort, reversed
    if (maxlines > 0 && n > maxlines) {
        std::stable_sort(keep.begin(), keep.end(), [&](int64_t a, int64_t b) { return S[a] < S[b]; });
        std::reverse(keep.begin(), keep.end());
        if ((int64_t)keep.size() > maxlines) keep.resize((size_t)maxlines);
    }
    std::stable_sort(keep.begin(), keep.end(), [&](int64_t a, int64_t b) { return nu[a] < nu[b]; });   // par.jl:188-191
    const int64_t L = (int64_t)keep.size();
    std::vector<double> o_nu(L), o_S(L), o_ga(L), o_gs(L), o_E(L), o_na(L), o_mu(L);
    std::vector<int16_t> o_iso(L);
    for (int64_t j = 0; j < L; j++) {
        const int64_t i = keep[j];
        if (M[i] != M_expected)   // SpectralLines holds one molecule (par.jl:239); the MOLPARAM rows handed in belong to M_expected
            return fail(CS_EINVAL, "record %lld is molecule %d, expected %d: SpectralLines objects must contain only one molecule's lines",
                        (long long)i, (int)M[i], M_expected);
        const int ii = iso_index(I[i]);
        if (ii < 1 || ii > niso) return fail(CS_EINVAL, "isotopologue '%c' of record %lld has no MOLPARAM row (%d rows)", I[i], (long long)i, niso);
        o_nu[j] = nu[i]; o_S[j] = S[i]; o_ga[j] = ga[i]; o_gs[j] = gs[i]; o_E[j] = Epp[i]; o_na[j] = na[i];
        o_iso[j] = (int16_t)ii;
        o_mu[j] = mu_table[ii - 1];
    }
    if (L_out) *L_out = L;
    return cs_gas_upload(ctx, slot, L, o_nu.data(), o_S.data(), o_ga.data(), o_gs.data(), o_E.data(), o_na.data(), o_mu.data(), o_iso.data(),
                         niso, ncheb, cheb);
}

int cs_gas_fetch(cs_ctx *ctx, int slot, int64_t L, double *nu, double *S, double *gamma_a, double *gamma_s, double *Epp, double *na,
                 double *mu_iso, int16_t *iso)
{
    if (!ctx || slot < 0 || slot >= CS_MAX_GAS || !ctx->gas[slot].present) return fail(CS_EINVAL, "gas slot is empty");
    GasTable &G = ctx->gas[slot];
    if (L != G.L) return fail(CS_EINVAL, "slot holds %lld lines, caller expects %lld", (long long)G.L, (long long)L);
    HIPCHK(hipSetDevice(ctx->device));
    const size_t nb = (size_t)L * sizeof(double);
    if (nu) HIPCHK(hipMemcpy(nu, G.nu.p, nb, hipMemcpyDeviceToHost));
    if (S) HIPCHK(hipMemcpy(S, G.S.p, nb, hipMemcpyDeviceToHost));
    if (gamma_a) HIPCHK(hipMemcpy(gamma_a, G.ga.p, nb, hipMemcpyDeviceToHost));
    if (gamma_s) HIPCHK(hipMemcpy(gamma_s, G.gs.p, nb, hipMemcpyDeviceToHost));
    if (Epp) HIPCHK(hipMemcpy(Epp, G.Epp.p, nb, hipMemcpyDeviceToHost));
    if (na) HIPCHK(hipMemcpy(na, G.na.p, nb, hipMemcpyDeviceToHost));
    if (mu_iso) HIPCHK(hipMemcpy(mu_iso, G.mu.p, nb, hipMemcpyDeviceToHost));
    if (iso) HIPCHK(hipMemcpy(iso, G.iso.p, (size_t)L * sizeof(int16_t), hipMemcpyDeviceToHost));
    return CS_OK;
}

// ---- multi-GPU --------------------------------------------------------------------------------------------
namespace {
// Host threads that drive the contexts of cs_fluxes_discretized_multi, kept for the life of the process: a thread's first HIP call
// costs milliseconds (per-thread runtime state), which a thread spawned per call would pay on every radiate!.
class WorkerPool {
    struct Worker {
        std::mutex m;
        std::condition_variable cv;
        std::function<void()> job;
        bool ready = false, done = true, quit = false;
        std::thread th;
    };
    std::vector<std::unique_ptr<Worker>> w_;
    std::mutex run_m_;
    static void loop(Worker *w)
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return w->ready || w->quit; });
            if (w->quit) return;
            w->ready = false;
            std::function<void()> job = std::move(w->job);
            lk.unlock();
            job();
            lk.lock();
            w->done = true;
            w->cv.notify_all();
        }
    }
public:
    // f(0) on the calling thread, f(1) .. f(n-1) on workers; returns when all are through
    void run(int n, const std::function<void(int)> &f)
    {
        std::lock_guard<std::mutex> g(run_m_);
        while ((int)w_.size() < n - 1) {   // (parked on their condition variables between calls; joined by stop())
            w_.emplace_back(new Worker());
            w_.back()->th = std::thread(loop, w_.back().get());
        }
        for (int i = 1; i < n; i++) {
            Worker *w = w_[i - 1].get();
            std::lock_guard<std::mutex> lk(w->m);
            w->job = [&f, i] { f(i); };
            w->done = false;
            w->ready = true;
            w->cv.notify_all();
        }
        f(0);
        for (int i = 1; i < n; i++) {
            Worker *w = w_[i - 1].get();
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return w->done; });
        }
    }
    // end and join every worker (cs_destroy of the process's last context: a long-lived host session does not keep parked threads
    // that have touched the GPU; the next multi-context call starts new ones)
    void stop()
    {
        std::lock_guard<std::mutex> g(run_m_);
        for (auto &w : w_) {
            { std::lock_guard<std::mutex> lk(w->m); w->quit = true; w->cv.notify_all(); }
            if (w->th.joinable()) w->th.join();
        }
        w_.clear();
    }
};
WorkerPool &worker_pool()
{
    static WorkerPool *p = new WorkerPool();   // (never destroyed itself: no static-destruction order to get wrong)
    return *p;
}
}  // namespace
static void pool_stop() { worker_pool().stop(); }

// cost per wavenumber, as a running sum: a fixed part (flux sweeps, interpolation carry, setup) + per gas its local line density rho
// [lines per cm^-1 within +-2 cm^-1] x (0.011 + 1.1e-6 nu) -- near-line pairs grow with the Doppler width.  The constants were fitted
// to the kernel times of the eight 1/8 shards of BASELINE configs[2] (profiles/r02_notes.md, r03_notes.md); a column whose balance
// they miss is re-cut from measured times (cs_rebalance_ranges).
static void model_cost_sum(int64_t nnu, const double *nu, int ngas, const int64_t *nlines, const double *const *line_nu, std::vector<double> &cum)
{
    cum.assign((size_t)nnu + 1, 0.0);
    std::vector<int64_t> lo(ngas, 0), hi(ngas, 0);   // nu ascends (asserted by the callers of the product path; any order still works,
    for (int64_t i = 0; i < nnu; i++) {              // the two cursors just move both ways): lines in [nu - 2, nu + 2] by two cursors per gas
        double w = 0.19;
        for (int gq = 0; gq < ngas; gq++) {
            const double *t = line_nu[gq];
            const int64_t L = nlines[gq];
            int64_t &a = lo[gq], &b = hi[gq];
            while (a < L && t[a] < nu[i] - 2.0) a++;
            while (a > 0 && t[a - 1] >= nu[i] - 2.0) a--;
            while (b < L && t[b] <= nu[i] + 2.0) b++;
            while (b > 0 && t[b - 1] > nu[i] + 2.0) b--;
            w += (double)(b - a) / 4.0 * (0.011 + 1.1e-6 * nu[i]);
        }
        cum[i + 1] = cum[i] + w;
    }
}
// nparts contiguous non-empty ranges of equal cost from the running cost sum; edges on multiples of 64 points where the grid allows
static int cut_equal_cost(int64_t nnu, const std::vector<double> &cum, int nparts, int64_t *ranges)
{
    std::vector<int64_t> edge(nparts + 1);
    const bool tiles = nnu >= (int64_t)64 * 4 * nparts;   // range edges on multiples of 64 points (the kernels' tile) where the grid allows
    for (int r = 0; r <= nparts; r++) {
        int64_t e = std::lower_bound(cum.begin(), cum.end(), cum[nnu] * r / nparts) - cum.begin();
        if (tiles) e = (e + 32) / 64 * 64;
        edge[r] = e;
    }
    edge[0] = 0;
    edge[nparts] = nnu;
    // every part keeps at least one point (one tile where edges sit on tiles): push edges up from the left, then down from the right
    const int64_t step = tiles ? 64 : 1;
    for (int r = 1; r < nparts; r++) edge[r] = std::max(edge[r], edge[r - 1] + step);
    for (int r = nparts - 1; r >= 1; r--) edge[r] = std::min(edge[r], edge[r + 1] - (r == nparts - 1 ? 1 : step));
    for (int r = 1; r < nparts; r++) edge[r] = std::max(edge[r], edge[r - 1] + 1);
    for (int r = 0; r < nparts; r++) {
        if (!(edge[r] >= 0 && edge[r] < edge[r + 1] && edge[r + 1] <= nnu)) return fail(CS_EINVAL, "could not cut %lld wavenumbers into %d non-empty ranges", (long long)nnu, nparts);
        ranges[2 * r] = edge[r];
        ranges[2 * r + 1] = edge[r + 1];
    }
    return CS_OK;
}

int cs_balanced_ranges(int64_t nnu, const double *nu, int ngas, const int64_t *nlines, const double *const *line_nu, int nparts, int64_t *ranges)
{
    if (!nu || nnu < 1 || nparts < 1 || !ranges || ngas < 0 || (ngas > 0 && (!nlines || !line_nu))) return fail(CS_EINVAL, "bad arguments");
    if (nparts > nnu) return fail(CS_EINVAL, "more parts (%d) than wavenumbers (%lld)", nparts, (long long)nnu);
    std::vector<double> cum;
    model_cost_sum(nnu, nu, ngas, nlines, line_nu, cum);
    return cut_equal_cost(nnu, cum, nparts, ranges);
}

int cs_rebalance_ranges(int64_t nnu, const double *nu, int ngas, const int64_t *nlines, const double *const *line_nu, int nparts,
                        const int64_t *prev_ranges, const double *prev_time, double fixed_time, int64_t *ranges)
{
    if (!nu || nnu < 1 || nparts < 1 || !ranges || !prev_ranges || !prev_time || ngas < 0 || (ngas > 0 && (!nlines || !line_nu)))
        return fail(CS_EINVAL, "bad arguments");
    if (nparts > nnu) return fail(CS_EINVAL, "more parts (%d) than wavenumbers (%lld)", nparts, (long long)nnu);
    for (int r = 0; r < nparts; r++) {
        const bool ok = prev_ranges[2 * r] == (r ? prev_ranges[2 * r - 1] : 0) && prev_ranges[2 * r + 1] > prev_ranges[2 * r] &&
                        prev_time[r] > 0.0 && std::isfinite(prev_time[r]);
        if (!ok || (r == nparts - 1 && prev_ranges[2 * r + 1] != nnu))
            return fail(CS_EINVAL, "prev_ranges must be a partition of the grid into %d non-empty ranges with positive measured times", nparts);
    }
    if (!(fixed_time >= 0.0)) return fail(CS_EINVAL, "fixed_time must be >= 0");
    std::vector<double> cum, cal((size_t)nnu + 1, 0.0);
    model_cost_sum(nnu, nu, ngas, nlines, line_nu, cum);
    // the model's density, rescaled part by part so that it reproduces what the part took beyond the share that does not move with the
    // edges; a part the model under-rates gets denser and will shrink
    for (int r = 0; r < nparts; r++) {
        const int64_t a = prev_ranges[2 * r], b = prev_ranges[2 * r + 1];
        const double t = std::max(prev_time[r] - fixed_time, 0.05 * prev_time[r]);
        const double sc = t / std::max(cum[b] - cum[a], 1e-300);
        for (int64_t i = a; i < b; i++) cal[i + 1] = cal[i] + (cum[i + 1] - cum[i]) * sc;
    }
    return cut_equal_cost(nnu, cal, nparts, ranges);
}

int cs_fluxes_discretized_multi(cs_ctx *const *ctxs, int nctx, int64_t nnu, const double *nu, int np, const double *P, double g, int nlobatto,
                                const double *T_nodes, const double *mu_nodes, const double *T_levels, int ngas, const int *gas_slots,
                                const int *shapes, const double *dnu_cuts, const double *conc, double sigma_gray, const double *sigma_extra,
                                const double *S_toa, const double *albedo, double theta_s, int nstream, double *tau, double *Mup,
                                double *Mdn, double *Fup, double *Fdn)
{
    if (!ctxs || nctx < 1) return fail(CS_EINVAL, "no contexts");
    for (int i = 0; i < nctx; i++)
        if (!ctxs[i]) return fail(CS_EINVAL, "context %d is NULL", i);
    if (!nu || !P || nnu < 1 || np < 2 || !Fup || !Fdn) return fail(CS_EINVAL, "bad grid arguments");
    if (nctx == 1)
        return cs_fluxes_discretized(ctxs[0], nnu, nu, np, P, g, nlobatto, T_nodes, mu_nodes, T_levels, ngas, gas_slots, shapes, dnu_cuts, conc,
                                     sigma_gray, sigma_extra, S_toa, albedo, theta_s, nstream, tau, Mup, Mdn, Fup, Fdn);
    int rc;
    if ((rc = check_ascending(nu, nnu))) return rc;
    if (nlobatto < 2 || nlobatto > CS_MAX_LOBATTO) return fail(CS_EINVAL, "nlobatto must be in [2,%d]", CS_MAX_LOBATTO);
    // the partition: equal estimated device time per context, from the line tables the first context holds (every context must hold
    // the same tables in the same slots)
    std::vector<int64_t> nl_(ngas), ranges(2 * (size_t)nctx);
    std::vector<const double *> ln_(ngas);
    for (int gi = 0; gi < ngas; gi++) {
        const int sl = gas_slots[gi];
        if (sl < 0 || sl >= CS_MAX_GAS) return fail(CS_EINVAL, "gas slot %d out of range", sl);
        for (int i = 0; i < nctx; i++)
            if (!ctxs[i]->gas[sl].present || ctxs[i]->gas[sl].L != ctxs[0]->gas[sl].L)
                return fail(CS_EINVAL, "gas slot %d must hold the same table on every context (context %d differs)", sl, i);
        nl_[gi] = ctxs[0]->gas[sl].L;
        ln_[gi] = ctxs[0]->gas[sl].h_nu.data();
    }
    // (partition and weights of an unchanged grid and gas line-up are kept by the first context: radiate! calls this once per time step)
    MultiPlan &mp = ctxs[0]->mplan;
    std::vector<uint64_t> gens(ngas);
    for (int gi = 0; gi < ngas; gi++) gens[gi] = ctxs[0]->gas[gas_slots[gi]].generation;
    bool calibrate = false;
    if (!(mp.nctx == nctx && (int64_t)mp.nu.size() == nnu && mp.gens == gens && memcmp(mp.nu.data(), nu, (size_t)nnu * sizeof(double)) == 0)) {
        if ((rc = cs_balanced_ranges(nnu, nu, ngas, nl_.data(), ln_.data(), nctx, ranges.data()))) return rc;
        // a new plan: once, inside this call, the model's partition is re-cut from what its ranges are measured to take on THIS column
        // (cs_rebalance_ranges) -- where every context has a device to itself, so that a range's time is its own (cs_set_tuning key 15 |
        // 32 on the first context: also on shared devices, for tests)
        calibrate = ngas > 0;
        for (int i = 0; i < nctx && calibrate; i++)
            for (int q = 0; q < i; q++)
                if (ctxs[q]->device == ctxs[i]->device) calibrate = false;
        if (ctxs[0]->tune[15] & 32) calibrate = ngas > 0;
        mp.nctx = nctx;
        mp.nu.assign(nu, nu + nnu);
        mp.gens = gens;
        mp.ranges = ranges;
        mp.wt.resize(nnu);   // trapezoid weights of the WHOLE grid (util.jl:26-33): shards use slices, so their band fluxes simply add
        for (int64_t j = 0; j < nnu; j++) mp.wt[j] = ((j > 0 ? nu[j] - nu[j - 1] : 0.0) + (j + 1 < nnu ? nu[j + 1] - nu[j] : 0.0)) / 2;
    }
  for (int pass = 0; pass < 2; pass++) {
    ranges = mp.ranges;
    const std::vector<double> &wt = mp.wt;
    std::vector<double> run_ms(nctx, 0.0);
    const int K = (np - 1) * (nlobatto - 1) + 1, nl = np - 1;
    std::vector<int> rcs(nctx, CS_OK);
    std::vector<std::string> msgs(nctx);
    // Contexts that share a device (several ranges per card) take turns at the kernels, in context order: range i + 1 starts when the
    // kernels of range i are through, so that the copy-back of one range -- tau, M+, M- are 24 bytes per spectral point over PCIe --
    // runs beside the kernels of the next instead of all ranges finishing, then copying, together.  Different devices run side by side.
    std::vector<int> prev(nctx, -1);
    for (int i = 0; i < nctx; i++)
        for (int q = 0; q < i; q++)
            if (ctxs[q]->device == ctxs[i]->device) prev[i] = q;
    std::mutex turn_m;
    std::condition_variable turn_cv;
    std::vector<char> kernels_done(nctx, 0);
    std::vector<double> Fpart((size_t)nctx * 2 * np, 0.0);
    // one host thread per context (persistent workers): its uploads, kernels and copy-backs run beside the others' (a context is not
    // re-entrant, but different contexts are independent; HIP calls carry their device through hipSetDevice per thread)
    auto work = [&](int i) {
        cs_ctx *ctx = ctxs[i];
        const int64_t a = ranges[2 * i], n = ranges[2 * i + 1] - a;
        std::vector<double> ex;
        if (sigma_extra) {   // [nnu, K] nu-fastest -> this shard's columns
            ex.resize((size_t)n * K);
            for (int k = 0; k < K; k++) std::copy(sigma_extra + (size_t)k * nnu + a, sigma_extra + (size_t)k * nnu + a + n, ex.begin() + (size_t)k * n);
        }
        const double gl = a > 0 ? nu[a - 1] : 0.0, gr = a + n < nnu ? nu[a + n] : 0.0;
        int r;
        if (column_matches(ctx, n, nu + a, np, P, g, nlobatto, ngas, gas_slots, shapes, dnu_cuts, sigma_gray, theta_s, nstream, tau != nullptr,
                           Mup || Mdn, nnu, a, gl, gr)) {
            Column &c = ctx->col;
            hipStream_t s = ctx->stream;
            r = hipSetDevice(ctx->device) == hipSuccess ? CS_OK : fail(CS_EHIP, "hipSetDevice(%d) failed", ctx->device);
            drop_graph(c);   // (spectra may have been switched on or off: other kernel arguments)
            c.has_extra = sigma_extra != nullptr;
            c.has_S = S_toa != nullptr && std::any_of(S_toa + a, S_toa + a + n, [](double x) { return x != 0.0; });
            c.has_alb = albedo != nullptr && std::any_of(albedo + a, albedo + a + n, [](double x) { return x != 0.0; });
            if (!r && c.has_extra) r = upload(c.extra, ex.data(), ex.size(), s);
            if (!r && c.has_S) r = upload(c.S_toa, S_toa + a, n, s);
            if (!r && c.has_alb) r = upload(c.albedo, albedo + a, n, s);
            if (!r) r = cs_column_update_state(ctx, T_nodes, mu_nodes, T_levels, conc, nullptr);
        } else {
            r = cs_column_setup(ctx, n, nu + a, wt.data() + a, np, P, g, nlobatto, T_nodes, mu_nodes, T_levels, ngas, gas_slots, shapes, dnu_cuts,
                                conc, sigma_gray, sigma_extra ? ex.data() : nullptr, S_toa ? S_toa + a : nullptr, albedo ? albedo + a : nullptr,
                                theta_s, nstream, tau != nullptr, (Mup || Mdn) ? 1 : 0);
            if (!r) { ctx->col.g_nnu = nnu; ctx->col.g_start = a; ctx->col.g_left = gl; ctx->col.g_right = gr; }
        }
        if (prev[i] >= 0) {
            std::unique_lock<std::mutex> lk(turn_m);
            turn_cv.wait(lk, [&] { return kernels_done[prev[i]] != 0; });
        }
        if (!r) r = cs_column_run(ctx, nullptr);
        if (!r && hipStreamSynchronize(ctx->stream) != hipSuccess) r = fail(CS_EHIP, "hipStreamSynchronize failed");
        if (!r && calibrate && pass == 0) {   // the range's own time: a second, warm evaluation (the first one loaded code objects)
            const auto t0 = std::chrono::steady_clock::now();
            r = cs_column_run(ctx, nullptr);
            if (!r && hipStreamSynchronize(ctx->stream) != hipSuccess) r = fail(CS_EHIP, "hipStreamSynchronize failed");
            run_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        {
            std::lock_guard<std::mutex> lk(turn_m);
            kernels_done[i] = 1;   // (also on failure: nobody is left waiting)
        }
        turn_cv.notify_all();
        if (!r) r = cs_column_fetch(ctx, n, np, tau ? tau + (size_t)a * nl : nullptr, Mup ? Mup + (size_t)a * np : nullptr,
                                    Mdn ? Mdn + (size_t)a * np : nullptr, Fpart.data() + (size_t)i * 2 * np, Fpart.data() + (size_t)i * 2 * np + np);
        rcs[i] = r;
        if (r) msgs[i] = g_err;   // (thread-local: carry it to the caller's thread)
    };
    worker_pool().run(nctx, work);
    for (int i = 0; i < nctx; i++)
        if (rcs[i]) return fail(rcs[i], "context %d (device %d): %s", i, ctxs[i]->device, msgs[i].c_str());
    if (calibrate && pass == 0) {
        calibrate = false;
        std::vector<int64_t> re(2 * (size_t)nctx);
        const double fixed = 0.3 * *std::min_element(run_ms.begin(), run_ms.end());   // (the launch chain of a step: does not move with the edges)
        if (cs_rebalance_ranges(nnu, nu, ngas, nl_.data(), ln_.data(), nctx, mp.ranges.data(), run_ms.data(), fixed, re.data()) == CS_OK && re != mp.ranges) {
            mp.ranges = re;
            continue;   // the call's results come from the re-cut partition, like every later call's
        }
    }
    for (int l = 0; l < np; l++) {   // fixed-order host sum: bitwise repeatable (SURVEY 8e's deterministic alternative to an all-reduce)
        double u = 0.0, d = 0.0;
        for (int i = 0; i < nctx; i++) { u += Fpart[(size_t)i * 2 * np + l]; d += Fpart[(size_t)i * 2 * np + np + l]; }
        Fup[l] = u;
        Fdn[l] = d;
    }
    return CS_OK;
  }
    return CS_OK;
}

int cs_streamnodes(int n, double *m, double *W)
{
    if (n < 1 || n > CS_MAX_STREAM) return fail(CS_EINVAL, "nstream must be in [1,%d]", CS_MAX_STREAM);
    double x[CS_MAX_STREAM], w[CS_MAX_STREAM];
    gauss_legendre(n, x, w);
    for (int i = 0; i < n; i++) {  // core/shared.jl:10-19
        const double th = (M_PI / 2) * (x[i] + 1) / 2;
        const double wi = (M_PI / 2) * w[i] / 2;
        m[i] = 1 / std::cos(th);
        W[i] = 2 * M_PI * wi * std::cos(th) * std::sin(th);
    }
    return CS_OK;
}

int cs_lobattonodes(int n, double *xs, double *ws)
{
    if (n < 2 || n > CS_MAX_LOBATTO) return fail(CS_EINVAL, "nlobatto must be in [2,%d]", CS_MAX_LOBATTO);
    double x[CS_MAX_LOBATTO], w[CS_MAX_LOBATTO];
    gauss_lobatto(n, x, w);
    for (int i = 0; i < n; i++) {  // core/discretized.jl:6-7
        xs[i] = (x[i] + 1) / 2;
        ws[i] = w[i] / 2;
    }
    return CS_OK;
}

int cs_devfn_batch(cs_ctx *ctx, int which, int64_t n, const double *x, const double *y, const double *z, double *out)
{
    if (!ctx || n < 0 || which < 0 || which > 2 || !x || !out) return fail(CS_EINVAL, "bad arguments");
    if ((which >= 1 && !y) || (which == 2 && !z)) return fail(CS_EINVAL, "this function takes more arguments");
    if (n == 0) return CS_OK;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DevBuf dx, dy, dz, dout;
    int rc;
    if ((rc = upload(dx, x, n, s)) || (y && (rc = upload(dy, y, n, s))) || (z && (rc = upload(dz, z, n, s)))) return rc;
    HIPCHK(dout.reserve(n * sizeof(double)));
    CS_LAUNCH(k_devfn, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, which, n, dx.as<double>(), dy.as<double>(), dz.as<double>(), dout.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return CS_OK;
}

int cs_faddeeva_batch(cs_ctx *ctx, int64_t n, const double *x, const double *y, double *out)
{
    if (!ctx || n < 0) return fail(CS_EINVAL, "bad arguments");
    if (n == 0) return CS_OK;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    int rc;
    if ((rc = upload(ctx->tmpA, x, n, s)) || (rc = upload(ctx->tmpB, y, n, s))) return rc;
    HIPCHK(ctx->tmpC.reserve(n * sizeof(double)));
    CS_LAUNCH(k_faddeeva, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, ctx->tmpA.as<double>(),
                       ctx->tmpB.as<double>(), ctx->tmpC.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, ctx->tmpC.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return CS_OK;
}

}  // extern "C"
