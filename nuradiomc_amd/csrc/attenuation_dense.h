// attenuation_dense.h -- the attenuation quadrature with dense (ray, frequency) packing and LDS-resident QUADPACK lists.
// Included by attenuation.hip inside namespace nrhip (it uses that file's GK rule, DQPSRT / DQELG restatements and models).
//
// What changes against attenuation_group_kernel (same arithmetic per item, hence the same bits and evaluation counts):
//   * a wave carries THREE groups of lanes instead of two: with 25 frequencies, two whole rays and half of a third one
//     (62..63 of 64 lanes carry an item instead of 50); two wave types alternate, the second takes the other half of the
//     split ray and two more rays, so five rays occupy two waves.  A group is the unit that shares Gauss-Kronrod node
//     records; all synchronisation stays inside the wave.
//   * the frequency-independent node records of both intervals of a bisection are produced by 63 lanes in one pass per
//     interval (lane = (group, node)), instead of 21 of 32 lanes per ray.
//   * no scratch at all.  The intervals QUADPACK creates are the nodes of a binary tree whose end points do not depend on the
//     frequency (b1 = (a + b) / 2 of the parent): ONE tree per group in LDS (end points, depth, child link); per lane the
//     slot results / errors (rlist, elist) in LDS columns, the slot -> tree-node map and the order list packed into two
//     64-bit registers (12 x 5 and 12 x 4 bits), the epsilon table in registers (DQELG unrolled per table length).  The frequencies of a ray may bisect in different order (3.7 % of the rays of the survey do): each lane
//     follows its own order on the shared tree; when the lanes of a group ask for different intervals in one round, the
//     group evaluates them one after the other.
//   * the 20 node values a rule needs twice stay in registers (the lists no longer compete for them).
// Capacity: ATTD_K list slots per item and ATTD_T tree nodes per group (the survey needs 12 / 24).  A group that would
// exceed either, or needs the general exp() range, is NOT finished here: its ray goes onto the overflow list and
// attenuation_group_kernel (any depth, limit 50) integrates it afterwards.
#pragma once

#define ATTD_K 12
#define ATTD_T 28
#define ATTD_NG 3

struct DenseMap {             // how the 64 lanes of a wave are dealt to (ray, frequency) items; wave types 0 / 1 alternate
    int rays_per_pair;        // rays taken by a type-0 + type-1 pair of waves
    int n_groups[2];
    int ray_off[2][ATTD_NG];  // ray of the group, relative to the pair's first ray
    int f0[2][ATTD_NG], nf[2][ATTD_NG], lane0[2][ATTD_NG];
};

// rays per wave pair and lane layout for n_freq frequencies (n_freq <= 32)
static inline DenseMap make_dense_map(int F)
{
    DenseMap d = {};
    auto set = [&](int t, int g, int ray, int f0, int nf, int lane0) {
        d.ray_off[t][g] = ray; d.f0[t][g] = f0; d.nf[t][g] = nf; d.lane0[t][g] = lane0;
    };
    if (3 * F <= 63) {            // three whole rays per wave
        d.rays_per_pair = 6;
        for (int t = 0; t < 2; t++) {
            d.n_groups[t] = 3;
            for (int g = 0; g < 3; g++) set(t, g, 3 * t + g, 0, F, g * F);
        }
    } else if (2 * F + (F + 1) / 2 <= 63) {   // two rays and half of a third one (F = 22 .. 25)
        const int h = (F + 1) / 2;
        d.rays_per_pair = 5;
        d.n_groups[0] = d.n_groups[1] = 3;
        set(0, 0, 0, 0, F, 0); set(0, 1, 1, 0, F, F); set(0, 2, 2, 0, h, 2 * F);
        set(1, 0, 2, h, F - h, 0); set(1, 1, 3, 0, F, F - h); set(1, 2, 4, 0, F, 2 * F - h);
    } else {                      // two whole rays per wave
        d.rays_per_pair = 4;
        for (int t = 0; t < 2; t++) {
            d.n_groups[t] = 2;
            for (int g = 0; g < 2; g++) set(t, g, 2 * t + g, 0, F, g * F);
        }
    }
    return d;
}

template <int MODEL> struct DenseRec { double ds, z; };
template <> struct DenseRec<1> { double ds, p[4]; };   // SP1: (a, b) of both frequency branches; the depth test is folded into ds

template <int MODEL>
struct DenseLds {
    DenseRec<MODEL> rec[ATTD_NG][2][21];
    double ta[ATTD_NG][ATTD_T], tb[ATTD_NG][ATTD_T];   // end points of the tree nodes
    double rlist[ATTD_K][64], elist[ATTD_K][64];
    unsigned char tchild[ATTD_NG][ATTD_T];             // first child (0: not bisected yet); the second one follows it
    unsigned char tdepth[ATTD_NG][ATTD_T];             // QAGP's level of the interval
    double exp_tab[64];                                // 2^(j / 64) for det_exp_tab_inrange (detmath.h)
};

// one 21-point rule from the group's node records; same operations and order as gk21_from_nodes, node values in registers.
// SP1: the caller has checked that every exp argument of the rule lies in [-700, 700]; and since every integrand value is
// >= +0 (ds >= 0 times min(exp, 1) > 0) the sum of |f| that QUADPACK keeps as resabs receives exactly the terms of resk in
// the same order from the same start: resabs == resk bit for bit (NaN alike), so it is not accumulated separately.
template <int MODEL>
__device__ __forceinline__ GK dense_rule(const DenseRec<MODEL>* __restrict__ r, double a, double b, const AttLane& lane, int sel, const double* exp_tab)
{
    const double WGK[11] = {
        0.011694638867371874278064396062192, 0.032558162307964727478818972459390,
        0.054755896574351996031381300244580, 0.075039674810919952767043140916190,
        0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
        0.123491976262065851077958109585166, 0.134709217311473325928054001771707,
        0.142775938577060080797094273138717, 0.147739104901338491374841515972068,
        0.149445554002916905664936468389821};
    const double WG[5] = {
        0.066671344308688137593568809893332, 0.149451349150580593145776339657697,
        0.219086362515982043995534934228163, 0.269266719309996355091226921569469,
        0.295524224714752870173815619188769};
    const double epmach = 2.220446049250313e-16, uflow = 2.2250738585072014e-308;
    auto fval = [&](int n) -> double {
        if constexpr (MODEL == 1) {
            const double x = r[n].p[sel] + r[n].p[sel + 1] * lane.w;
            return r[n].ds * fmin(det_exp_tab_inrange(x, exp_tab), 1.);
        } else {
            return r[n].ds / attenuation_length(r[n].z, lane);
        }
    };
    auto fval2 = [&](int n1, int n2, double& f1, double& f2) {
        if constexpr (MODEL == 1) {
            const double x1 = r[n1].p[sel] + r[n1].p[sel + 1] * lane.w, x2 = r[n2].p[sel] + r[n2].p[sel + 1] * lane.w;
            double e1, e2;
            det_exp_tab_inrange2(x1, x2, exp_tab, e1, e2);
            f1 = r[n1].ds * fmin(e1, 1.);
            f2 = r[n2].ds * fmin(e2, 1.);
        } else {
            f1 = fval(n1);
            f2 = fval(n2);
        }
    };
    double fv1[10], fv2[10];
    const double hlgth = 0.5 * (b - a), dhlgth = fabs(hlgth);
    double resg = 0.;
    const double fc = fval(0);
    double resk = WGK[10] * fc;
    double resabs = fabs(resk);
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int jtw = 2 * j + 1;
        double f1, f2;
        fval2(1 + 2 * j, 2 + 2 * j, f1, f2);
        fv1[jtw] = f1; fv2[jtw] = f2;
        const double fsum = f1 + f2;
        resg += WG[j] * fsum;
        resk += WGK[jtw] * fsum;
        if constexpr (MODEL != 1) resabs += WGK[jtw] * (fabs(f1) + fabs(f2));
        __builtin_amdgcn_sched_barrier(0);   // one pair of nodes at a time: the 20 kept values leave no room for more in flight
    }
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int jtwm1 = 2 * j;
        double f1, f2;
        fval2(11 + 2 * j, 12 + 2 * j, f1, f2);
        fv1[jtwm1] = f1; fv2[jtwm1] = f2;
        const double fsum = f1 + f2;
        resk += WGK[jtwm1] * fsum;
        if constexpr (MODEL != 1) resabs += WGK[jtwm1] * (fabs(f1) + fabs(f2));
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MODEL == 1) resabs = resk;
    const double reskh = resk * 0.5;
    double resasc = WGK[10] * fabs(fc - reskh);
#pragma unroll
    for (int j = 0; j < 10; j++) resasc += WGK[j] * (fabs(fv1[j] - reskh) + fabs(fv2[j] - reskh));
    GK o;
    o.result = resk * hlgth;
    o.resabs = resabs * dhlgth;
    o.resasc = resasc * dhlgth;
    o.abserr = fabs((resk - resg) * hlgth);
    if (o.resasc != 0. && o.abserr != 0.) {
        const double q = 200. * o.abserr / o.resasc;
        o.abserr = o.resasc * fmin(1., q * sqrt(q));
    }
    if (o.resabs > uflow / (50. * epmach)) o.abserr = fmax((epmach * 50.) * o.resabs, o.abserr);
    return o;
}

// DQPSRT on the lane's LDS columns (1-based indices as in sort_errors; el(i) = elist[i], io(i) = iord[i])
template <class EL, class IOG, class IOS>
__device__ __forceinline__ void dense_sort_errors(int last, int& maxerr, double& ermax, EL&& el, IOG&& io, IOS&& ios, int& nrmax)
{
    const int limit = QLIM;
    if (last <= 2) {
        ios(1, 1);
        ios(2, 2);
    } else {
        const double errmax = el(maxerr);
        if (nrmax != 1) {
            const int ido = nrmax - 1;
            for (int i = 1; i <= ido; i++) {
                const int isucc = io(nrmax - 1);
                if (errmax <= el(isucc)) break;
                ios(nrmax, isucc);
                nrmax--;
            }
        }
        int jupbn = last;
        if (last > (limit / 2 + 2)) jupbn = limit + 3 - last;
        const double errmin = el(last);
        const int jbnd = jupbn - 1;
        int i = nrmax + 1;
        bool found = false;
        for (; i <= jbnd; i++) {
            const int isucc = io(i);
            if (errmax >= el(isucc)) { found = true; break; }
            ios(i - 1, isucc);
        }
        if (!found) {
            ios(jbnd, maxerr);
            ios(jupbn, last);
        } else {
            ios(i - 1, maxerr);
            int k = jbnd;
            bool placed = false;
            for (int j = i; j <= jbnd; j++) {
                const int isucc = io(k);
                if (errmin < el(isucc)) {
                    ios(k + 1, last);
                    placed = true;
                    break;
                }
                ios(k + 1, isucc);
                k--;
            }
            if (!placed) ios(i, last);
        }
    }
    maxerr = io(nrmax);
    ermax = el(maxerr);
}

#define ATTD_E 9   // entries of the epsilon table kept per item (numrl2 + 2 <= ATTD_E)

// The epsilon table rlist2[1 .. ATTD_E] as named scalars: registers cannot be indexed by data, and an array would end up in
// scratch as soon as the optimiser merges two stores at different constant indices into one store at a selected index.
struct EpsTab { double v1, v2, v3, v4, v5, v6, v7, v8, v9; };
template <int I> __device__ __forceinline__ double& eps_at(EpsTab& t)
{
    static_assert(I >= 1 && I <= ATTD_E, "epsilon table index");
    if constexpr (I == 1) return t.v1;
    else if constexpr (I == 2) return t.v2;
    else if constexpr (I == 3) return t.v3;
    else if constexpr (I == 4) return t.v4;
    else if constexpr (I == 5) return t.v5;
    else if constexpr (I == 6) return t.v6;
    else if constexpr (I == 7) return t.v7;
    else if constexpr (I == 8) return t.v8;
    else return t.v9;
}
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// DQELG (epsilon_extrap) for a table of exactly N entries on entry, every index a compile-time constant.  Same operations in the
// same order as epsilon_extrap.
template <int N>
__device__ __forceinline__ void dense_epsilon_static(EpsTab& e, int& n, double& result, double& abserr, double& r3a, double& r3b,
                                                     double& r3c, int& nres)
{
    const double epmach = 2.220446049250313e-16, oflow = 1.7976931348623157e+308;
    nres++;
    abserr = oflow;
    result = eps_at<N>(e);
    if constexpr (N >= 3) {
        eps_at<N + 2>(e) = eps_at<N>(e);
        constexpr int newelm = (N - 1) / 2;
        eps_at<N>(e) = oflow;
        bool live = true, early = false;
        n = N;
        static_for<1, newelm + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int k1 = N - 2 * (i - 1), k2 = k1 - 1, k3 = k1 - 2;
            if (live) {
                double res = eps_at<k1 + 2>(e);
                const double e0 = eps_at<k3>(e), e1 = eps_at<k2>(e), e2 = res;
                const double e1abs = fabs(e1);
                const double delta2 = e2 - e1, err2 = fabs(delta2), tol2 = fmax(fabs(e2), e1abs) * epmach;
                const double delta3 = e1 - e0, err3 = fabs(delta3), tol3 = fmax(e1abs, fabs(e0)) * epmach;
                if (!(err2 > tol2 || err3 > tol3)) {
                    result = res;
                    abserr = fmax(err2 + err3, 5. * epmach * fabs(result));
                    early = true;
                    live = false;
                } else {
                    const double e3 = eps_at<k1>(e);
                    eps_at<k1>(e) = e1;
                    const double delta1 = e1 - e3, err1 = fabs(delta1), tol1 = fmax(e1abs, fabs(e3)) * epmach;
                    if (err1 <= tol1 || err2 <= tol2 || err3 <= tol3) { n = i + i - 1; live = false; }
                    else {
                        const double ss = 1. / delta1 + 1. / delta2 - 1. / delta3;
                        const double epsinf = fabs(ss * e1);
                        if (!(epsinf > 1e-4)) { n = i + i - 1; live = false; }
                        else {
                            res = e1 + 1. / ss;
                            eps_at<k1>(e) = res;
                            const double error = err2 + fabs(res - e2) + err3;
                            if (!(error > abserr)) {
                                abserr = error;
                                result = res;
                            }
                        }
                    }
                }
            }
        });
        if (early) return;
        constexpr int ib0 = (N % 2 == 0) ? 2 : 1;
        static_for<0, newelm + 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            eps_at<ib0 + 2 * i>(e) = eps_at<ib0 + 2 * i + 2>(e);
        });
        if (n != N) {   // n = 1, 3, 5: the last n entries move to the front
            static_for<0, (N - 1) / 2>([&](auto jc) {
                constexpr int nn = 2 * decltype(jc)::value + 1;
                if constexpr (nn < N) {
                    if (n == nn)
                        static_for<1, nn + 1>([&](auto ic) {
                            constexpr int i = decltype(ic)::value;
                            eps_at<i>(e) = eps_at<N - nn + i>(e);
                        });
                }
            });
        }
        if (nres < 4) {
            if (nres == 1) r3a = result;
            else if (nres == 2) r3b = result;
            else r3c = result;
            abserr = oflow;
        } else {
            abserr = fabs(result - r3c) + fabs(result - r3b) + fabs(result - r3a);
            r3a = r3b;
            r3b = r3c;
            r3c = result;
        }
    }
    abserr = fmax(abserr, 5. * epmach * fabs(result));
}

template <int MODEL>
__global__ void __launch_bounds__(64, ATT_WAVES)
attenuation_dense_kernel(long n_rays, const double* __restrict__ C0, const double* __restrict__ zint, int n_freq,
                         const double* __restrict__ freqs, IceConst m, double* __restrict__ att, int* __restrict__ neval,
                         const int* __restrict__ ray_index, unsigned long long* __restrict__ eval_counter, DenseMap map,
                         int* __restrict__ overflow_count, int* __restrict__ overflow_list)
{
    __shared__ DenseLds<MODEL> L;
    L.exp_tab[threadIdx.x & 63] = det_exp_tab64[threadIdx.x & 63];   // (one wave per block: ordered before the first rule by the wave's own syncs)
    const double epmach = 2.220446049250313e-16, uflow = 2.2250738585072014e-308, oflow = 1.7976931348623157e+308;
    const double epsabs = 1.49e-8, epsrel = 1e-2;
    const int limit = QLIM;
    const int lane = threadIdx.x;
    const int wtype = blockIdx.x & 1;             // gridDim.x is even: a block keeps its type
    const int NG = map.n_groups[wtype];
    // ---- the lane's two roles -----------------------------------------------------------------------------------------
    int g = -1, jf = 0, glane0 = 0, gnf = 0;     // item role: group, frequency
#pragma unroll
    for (int q = 0; q < ATTD_NG; q++)
        if (q < NG && lane >= map.lane0[wtype][q] && lane < map.lane0[wtype][q] + map.nf[wtype][q]) {
            g = q; jf = map.f0[wtype][q] + lane - map.lane0[wtype][q]; glane0 = map.lane0[wtype][q]; gnf = map.nf[wtype][q];
        }
    const unsigned long long gmask = (g >= 0) ? (((1ULL << gnf) - 1ULL) << glane0) : 0ULL;
    const int tg = (lane < 21 * NG) ? lane / 21 : -1;   // node role: group and node of the records this lane computes
    const int tnode = lane - 21 * (tg < 0 ? 0 : tg);
    unsigned long long tgmask = 0ULL;
    int roff_g = -1, roff_t = -1;   // ray of the lane's group / task group, relative to the pair's first ray
#pragma unroll
    for (int q = 0; q < ATTD_NG; q++) {
        if (q == tg) { tgmask = ((1ULL << map.nf[wtype][q]) - 1ULL) << map.lane0[wtype][q]; roff_t = map.ray_off[wtype][q]; }
        if (q == g) roff_g = map.ray_off[wtype][q];
    }
    const int rays_per_pair = map.rays_per_pair;
    AttLane al;
    al.model = MODEL;
    al.f = (g >= 0 && jf < n_freq) ? freqs[jf] : 1.;
    al.w = det_log(al.f);
    const int sel = (MODEL == 1 && !(al.f < 1.)) ? 2 : 0;
    // SP1: the exp arguments x = a(z) + b(z) ln f of a node are within [-700, 700] for all frequencies if |a| + |b| max|ln f| is
    // (checked per node by the lane that makes the record, not per (node, frequency))
    double wmax = (g >= 0 && jf < n_freq) ? fabs(al.w) : 0.;
    for (int off = 32; off > 0; off >>= 1) wmax = fmax(wmax, __shfl_xor(wmax, off));
    const unsigned long long taskmask = (g >= 0) ? (((1ULL << 21) - 1ULL) << (21 * g)) : 0ULL;   // the lanes making the group's records
    const int gi = g < 0 ? 0 : g;     // safe index for LDS addressing of idle lanes
    const int tgi = tg < 0 ? 0 : tg;
    unsigned long long my_evals = 0;
#ifdef NRHIP_ATT_TIMING
    unsigned long long tacc[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    const unsigned long long t_kernel = __builtin_amdgcn_s_memtime();
#define DT_MARK(t) unsigned long long t = __builtin_amdgcn_s_memtime()
#define DT_ADD(i, t0) do { tacc[i] += __builtin_amdgcn_s_memtime() - (t0); } while (0)
#else
#define DT_MARK(t)
#define DT_ADD(i, t0)
#endif
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    const long n_pairs_total = (n_rays + rays_per_pair - 1) / rays_per_pair;
    const long pair_stride = gridDim.x >> 1;
    // the rays of a wave's groups; the NEXT pair's are requested before the current one's results are stored
    struct PairRays { long gray; double gC0, gz1, gz2m, gzt, tC0, tzt; };
    auto load_rays = [&](long pair) -> PairRays {
        // (per-lane addresses -- the lane's own group and its task group -- so that these are vector loads, waited for with
        // vmcnt: scalar loads share the LDS operations' counter and would be waited for at the next LDS access)
        PairRays r{-1, NAN, 0., 0., 0., NAN, 0.};
        if (pair >= n_pairs_total) return r;
        const long li = pair * rays_per_pair + roff_g, lt = pair * rays_per_pair + roff_t;
        if (roff_g >= 0 && li < n_rays) {
            const long ray = ray_index ? ray_index[li] : li;
            r.gray = ray; r.gC0 = C0[ray]; r.gz1 = zint[3 * ray]; r.gz2m = zint[3 * ray + 1]; r.gzt = zint[3 * ray + 2];
        }
        if (roff_t >= 0 && lt < n_rays) {
            const long ray = ray_index ? ray_index[lt] : lt;
            r.tC0 = C0[ray]; r.tzt = zint[3 * ray + 2];
        }
        return r;
    };
    PairRays nxt = load_rays(blockIdx.x >> 1);
    for (long pair = blockIdx.x >> 1; pair < n_pairs_total; pair += pair_stride) {
        DT_MARK(t_pair);
        const double gC0 = nxt.gC0, gz1 = nxt.gz1, gz2m = nxt.gz2m, gzt = nxt.gzt;
        const long gray = nxt.gray;
        const bool ray_ok = (g >= 0) && gray >= 0 && jf < n_freq;
        const bool valid = ray_ok && !isnan(gC0);
        const bool with_point = (gz1 < gzt && gzt < gz2m);
        AttItem tit;      // the node role's ray
        tit.C0 = nxt.tC0;
        tit.z_turn = nxt.tzt;
        tit.lane = al;
        tit.lane.model = MODEL;

        // ---- QUADPACK state of the lane (dqagse / dqagpe) ----------------------------------------------------------------
        EpsTab ep{0., 0., 0., 0., 0., 0., 0., 0., 0.};   // the epsilon table rlist2[1 .. ATTD_E]
        double r3a = 0., r3b = 0., r3c = 0.;
        unsigned long long iop = 0ULL, nip = 0ULL;   // iord[1 .. 12] (4 bits each) and the slot -> tree-node map (5 bits each)
        auto iord_at = [&](int i) -> int { return (int)((iop >> (4 * (i - 1))) & 15ULL); };
        auto iord_set = [&](int i, int v) {
            const int sh = 4 * (i - 1);
            iop = (iop & ~(15ULL << sh)) | ((unsigned long long)v << sh);
        };
        auto node_of = [&](int slot) -> int { return (int)((nip >> (5 * (slot - 1))) & 31ULL); };
        auto node_set = [&](int slot, int v) {
            const int sh = 5 * (slot - 1);
            nip = (nip & ~(31ULL << sh)) | ((unsigned long long)v << sh);
        };
        auto elist_at = [&](int i) -> double { return L.elist[i - 1][lane]; };
        auto width_of = [&](int slot) -> double { const int nn = node_of(slot); return fabs(L.tb[gi][nn] - L.ta[gi][nn]); };
        auto level_of = [&](int slot) -> int { return (int)L.tdepth[gi][node_of(slot)]; };
        double result = 0., abserr = 0., resabs = 0., errsum = 0., errbnd = 0., errmax = 0., area = 0.;
        double erlarg = 0., ertest = 0., correc = 0., small = 0.;
        int ier = 0, ierro = 0, iroff1 = 0, iroff2 = 0, iroff3 = 0, ksgn = 1, ktmin = 0, last = 0, maxerr = 1, neval_ = 0,
            nres = 0, nrmax = 1, numrl2 = 1, levmax = 1, levcur = 0;
        bool extrap = false, noext = false, busy = false, ovf = false, entered_loop = false;
        const bool qagp = with_point;
        int exit_code = 0;
        int tcount = 2;   // tree nodes of the lane's group (kept alike by all its lanes)
        GK g1, g2;
        g1.result = g1.abserr = g1.resabs = g1.resasc = 0.;
        g2 = g1;

        // roots of the group's tree: (a, b), or (lo, point) and (point, hi)
        if (g >= 0 && lane == glane0) {
            const double lo = fmin(gz1, gz2m), hi = fmax(gz1, gz2m);
            L.ta[gi][0] = qagp ? lo : gz1;
            L.tb[gi][0] = with_point ? gzt : gz2m;
            L.ta[gi][1] = gzt;
            L.tb[gi][1] = hi;
            L.tdepth[gi][0] = L.tdepth[gi][1] = 0;
            L.tchild[gi][0] = L.tchild[gi][1] = 0;
        }
        wave_sync();
        DT_ADD(9, t_pair);

        // Rounds of rule evaluations.  pend: the lane needs the estimates of the two children of tree node tn (tn < 0: of the
        // roots, in the first round; `two` false there for rays without an inner turning point).  Lanes of a group that ask for
        // different nodes are served one node after the other.  A round leaves g1, g2 and the first child in c_mine.
        bool first = true, pend = valid, two = with_point;
        int tn = -1, c_mine = 0;
        double erlast = 0.;
        while (true) {
            while (true) {
                const unsigned long long pm = __ballot(pend);
                if (pm == 0ULL) break;
                DT_MARK(t_r0);
                // -- item role: the node the group serves now, its children
                const unsigned long long mine = pm & gmask;
                const bool has = mine != 0ULL;
                const int leader = has ? (__ffsll((long long)mine) - 1) : lane;
                const int ltn = __shfl(tn, leader);
                int c = 0;
                bool grp_ovf = false;
                if (has && ltn >= 0) {
                    c = (int)L.tchild[gi][ltn];
                    if (c == 0) {
                        c = tcount;
                        if (c + 2 > ATTD_T) grp_ovf = true;
                        else {
                            if (lane == leader) {
                                const double pa_ = L.ta[gi][ltn], pb_ = L.tb[gi][ltn], mid = 0.5 * (pa_ + pb_);
                                const unsigned char dep = (unsigned char)(L.tdepth[gi][ltn] + 1);
                                L.ta[gi][c] = pa_; L.tb[gi][c] = mid;
                                L.ta[gi][c + 1] = mid; L.tb[gi][c + 1] = pb_;
                                L.tdepth[gi][c] = L.tdepth[gi][c + 1] = dep;
                                L.tchild[gi][c] = L.tchild[gi][c + 1] = 0;
                                L.tchild[gi][ltn] = (unsigned char)c;
                            }
                            tcount += 2;
                        }
                    }
                }
                if (grp_ovf) {          // every lane of the group sees it: the ray is left to the general kernel
                    ovf = true;
                    busy = false;
                    pend = false;
                }
                const bool my_turn = pend && tn == ltn;
                wave_sync();
                DT_ADD(0, t_r0);
                DT_MARK(t_r1);
                // -- node role: records of the served node's children for the lane's task group
                unsigned long long rng_bad = 0ULL;
                {
                    const unsigned long long tm = pm & tgmask;
                    const int tl = (tm != 0ULL) ? (__ffsll((long long)tm) - 1) : lane;
                    const int tc = __shfl(grp_ovf ? -1 : c, tl);
                    const int ttwo = __shfl((int)two, tl);
                    bool bad = false;
                    if (tm != 0ULL && tc >= 0) {
#pragma unroll
                        for (int iv = 0; iv < 2; iv++) {   // (both intervals in one block: their dependent chains interleave)
                            if (iv == 1 && !ttwo) break;
                            const double ia = L.ta[tgi][tc + iv], ib = L.tb[tgi][tc + iv];
                            NodeShared n1 = node_shared(gk_node(tnode, ia, ib), tit, m);
                            DenseRec<MODEL> rr;
                            if constexpr (MODEL == 1) {
                                // the integrand is ds * (z > 0 ? 0 : min(exp(x), 1)): the depth test is a property of the node
                                if (n1.z > 0) n1.ds = n1.ds * 0.;
                                rr.ds = n1.ds;
                                rr.p[0] = n1.p[0]; rr.p[1] = n1.p[1]; rr.p[2] = n1.p[2]; rr.p[3] = n1.p[3];
                                bad = bad || !(fabs(n1.p[0]) + fabs(n1.p[1]) * wmax <= 690.) || !(fabs(n1.p[2]) + fabs(n1.p[3]) * wmax <= 690.);
                            } else {
                                rr.ds = n1.ds;
                                rr.z = n1.z;
                            }
                            L.rec[tgi][iv][tnode] = rr;
                        }
                    }
                    rng_bad = __ballot(bad);
                }
                wave_sync();
                DT_ADD(1, t_r1);
                DT_MARK(t_r2);
                // -- item role: the lane's own sums over the shared records
                if (my_turn) {
                    c_mine = c;
                    const bool ok = (rng_bad & taskmask) == 0ULL;
                    if (ok) {
#pragma unroll 1
                        for (int iv = 0; iv < (two ? 2 : 1); iv++) {   // (one copy of the rule's code)
                            const GK q = dense_rule<MODEL>(&L.rec[gi][iv][0], L.ta[gi][c + iv], L.tb[gi][c + iv], al, sel, L.exp_tab);
                            if (iv) g2 = q;
                            else g1 = q;
                        }
                    } else {            // exp argument outside [-700, 700] (never on physical rays): the general rule
                        AttItem it;
                        it.C0 = gC0;
                        it.z_turn = gzt;
                        it.lane = al;
                        g1 = gk21(L.ta[gi][c], L.tb[gi][c], it, m);
                        if (two) g2 = gk21(L.ta[gi][c + 1], L.tb[gi][c + 1], it, m);
                    }
                    pend = false;
                }
                wave_sync();
                DT_ADD(2, t_r2);
            }
            if (first) {
                // ---- first estimate(s) --------------------------------------------------------------------------------------
                DT_MARK(t_init);
                if (valid && !ovf) {
                    if (qagp) {
                        double e1_ = g1.abserr, e2_ = g2.abserr;
                        abserr = (abserr + g1.abserr) + g2.abserr;
                        result = (result + g1.result) + g2.result;
                        const bool nd1 = (g1.abserr == g1.resasc && g1.abserr != 0.), nd2 = (g2.abserr == g2.resasc && g2.abserr != 0.);
                        resabs = (resabs + g1.resabs) + g2.resabs;
                        if (nd1) e1_ = abserr;
                        if (nd2) e2_ = abserr;
                        errsum = (errsum + e1_) + e2_;
                        L.elist[0][lane] = e1_; L.elist[1][lane] = e2_;
                        L.rlist[0][lane] = g1.result; L.rlist[1][lane] = g2.result;
                        node_set(1, 0); node_set(2, 1);
                        iord_set(1, 1); iord_set(2, 2);
                        last = 2;
                        neval_ = 42;
                        const double dres = fabs(result);
                        errbnd = fmax(epsabs, epsrel * dres);
                        if (abserr <= 100. * epmach * resabs && abserr > errbnd) ier = 2;
                        if (!(e1_ > e2_)) { iord_set(1, 2); iord_set(2, 1); }
                        if (!(ier != 0 || abserr <= errbnd)) {
                            ep.v1 = result;
                            maxerr = iord_at(1);
                            errmax = elist_at(maxerr);
                            area = result;
                            nrmax = 1;
                            numrl2 = 1;
                            erlarg = errsum;
                            ertest = errbnd;
                            abserr = oflow;
                            ksgn = (dres >= (1. - 50. * epmach) * resabs) ? 1 : -1;
                            last = 3;
                            busy = true;
                        }
                    } else {
                        result = g1.result;
                        abserr = g1.abserr;
                        const double defabs = g1.resabs;
                        const double dres = fabs(result);
                        errbnd = fmax(epsabs, epsrel * dres);
                        last = 1;
                        L.rlist[0][lane] = result;
                        L.elist[0][lane] = abserr;
                        node_set(1, 0);
                        iord_set(1, 1);
                        if (abserr <= 100. * epmach * defabs && abserr > errbnd) ier = 2;
                        if (ier != 0 || (abserr <= errbnd && abserr != g1.resasc) || abserr == 0.) {
                            neval_ = 21;
                        } else {
                            ep.v1 = result;
                            errmax = abserr;
                            maxerr = 1;
                            area = result;
                            errsum = abserr;
                            abserr = oflow;
                            nrmax = 1;
                            numrl2 = 2;
                            ksgn = (dres >= (1. - 50. * epmach) * defabs) ? 1 : -1;
                            resabs = defabs;
                            last = 2;
                            busy = true;
                        }
                    }
                }
                entered_loop = busy;
                first = false;
                DT_ADD(6, t_init);
            } else if (busy) {
                // ---- after a bisection --------------------------------------------------------------------------------------
                DT_MARK(t_book);
                do {
                    const int c = c_mine;
                    const double a1 = L.ta[gi][c], b1 = L.tb[gi][c], a2 = L.ta[gi][c + 1], b2 = L.tb[gi][c + 1];
                    neval_ += 42;
                    const double rmax = L.rlist[maxerr - 1][lane];
                    const double area12 = g1.result + g2.result;
                    const double erro12 = g1.abserr + g2.abserr;
                    errsum = errsum + erro12 - errmax;
                    area = area + area12 - rmax;
                    if (g1.resasc != g1.abserr && g2.resasc != g2.abserr) {
                        if (fabs(rmax - area12) <= 1e-5 * fabs(area12) && erro12 >= 0.99 * errmax) {
                            if (extrap) iroff2++;
                            else iroff1++;
                        }
                        if (last > 10 && erro12 > errmax) iroff3++;
                    }
                    errbnd = fmax(epsabs, epsrel * fabs(area));
                    if (iroff1 + iroff2 >= 10 || iroff3 >= 20) ier = 2;
                    if (iroff2 >= 5) ierro = 3;
                    if (last == limit) ier = 1;
                    if (fmax(fabs(a1), fabs(b2)) <= (1. + 100. * epmach) * (fabs(a2) + 1000. * uflow)) ier = 4;
                    if (g2.abserr > g1.abserr) {
                        node_set(maxerr, c + 1);
                        node_set(last, c);
                        L.rlist[maxerr - 1][lane] = g2.result;
                        L.rlist[last - 1][lane] = g1.result;
                        L.elist[maxerr - 1][lane] = g2.abserr;
                        L.elist[last - 1][lane] = g1.abserr;
                    } else {
                        node_set(maxerr, c);
                        node_set(last, c + 1);
                        L.rlist[maxerr - 1][lane] = g1.result;
                        L.rlist[last - 1][lane] = g2.result;
                        L.elist[maxerr - 1][lane] = g1.abserr;
                        L.elist[last - 1][lane] = g2.abserr;
                    }
                    DT_ADD(4, t_book);
                    DT_MARK(t_sort);
                    dense_sort_errors(last, maxerr, errmax, elist_at, iord_at, iord_set, nrmax);
                    DT_ADD(5, t_sort);
                    if (errsum <= errbnd) { exit_code = 1; busy = false; break; }
                    if (ier != 0) { exit_code = 2; busy = false; break; }
                    if (!qagp && last == 2) {
                        small = fabs(gz2m - gz1) * 0.375;
                        erlarg = errsum;
                        ertest = errbnd;
                        ep.v2 = area;
                        break;  // continue
                    }
                    if (noext) break;  // continue
                    erlarg -= erlast;
                    if (qagp) { if (levcur + 1 <= levmax) erlarg += erro12; }
                    else      { if (fabs(b1 - a1) > small) erlarg += erro12; }
                    if (!extrap) {
                        const bool is_smallest = qagp ? !(level_of(maxerr) + 1 <= levmax) : !(width_of(maxerr) > small);
                        if (!is_smallest) break;  // continue
                        extrap = true;
                        nrmax = 2;
                    }
                    if (!(ierro == 3 || erlarg <= ertest)) {
                        int jupbnd = last;
                        if (last > (2 + limit / 2)) jupbnd = limit + 3 - last;
                        bool cont = false;
                        for (int k = nrmax; k <= jupbnd; k++) {
                            maxerr = iord_at(nrmax);
                            errmax = elist_at(maxerr);
                            const bool big = qagp ? (level_of(maxerr) + 1 <= levmax) : (width_of(maxerr) > small);
                            if (big) { cont = true; break; }
                            nrmax++;
                        }
                        if (cont) break;  // continue
                    }
                    DT_MARK(t_eps);
                    numrl2++;
                    if (numrl2 + 2 > ATTD_E) { ovf = true; busy = false; break; }   // (the group's ray goes to the general kernel)
                    double reseps = 0., abseps = 0.;
                    const bool skip_eps = qagp && numrl2 <= 2;
                    // rlist2[numrl2] = area and DQELG, unrolled per table length (registers cannot be indexed by data)
                    // (the empty asm keeps the compiler from merging the cases' stores into one store at a data-dependent index,
                    // which would put the whole table into scratch)
#define ATTD_EPS_CASE(NN)                                                                                             \
    {                                                                                                                 \
        double v_ = area;                                                                                             \
        asm volatile("; epsilon table, %0 entries" : "+v"(v_) : "n"(NN));                                             \
        eps_at<NN>(ep) = v_;                                                                                                  \
        if (!skip_eps) dense_epsilon_static<NN>(ep, numrl2, reseps, abseps, r3a, r3b, r3c, nres);                      \
    }
                    if (numrl2 == 1) ATTD_EPS_CASE(1)
                    else if (numrl2 == 2) ATTD_EPS_CASE(2)
                    else if (numrl2 == 3) ATTD_EPS_CASE(3)
                    else if (numrl2 == 4) ATTD_EPS_CASE(4)
                    else if (numrl2 == 5) ATTD_EPS_CASE(5)
                    else if (numrl2 == 6) ATTD_EPS_CASE(6)
                    else ATTD_EPS_CASE(7)
#undef ATTD_EPS_CASE
                    DT_ADD(11, t_eps);
                    if (!skip_eps) {
                        ktmin++;
                        if (ktmin > 5 && abserr < 1e-3 * errsum) ier = 5;
                        if (abseps < abserr) {
                            ktmin = 0;
                            abserr = abseps;
                            result = reseps;
                            correc = erlarg;
                            ertest = fmax(epsabs, epsrel * fabs(reseps));
                            if (qagp ? (abserr < ertest) : (abserr <= ertest)) { exit_code = 2; busy = false; break; }
                        }
                        if (numrl2 == 1) noext = true;
                        if (qagp ? (ier >= 5) : (ier == 5)) { exit_code = 2; busy = false; break; }
                    }
                    maxerr = iord_at(1);
                    errmax = elist_at(maxerr);
                    nrmax = 1;
                    extrap = false;
                    if (qagp) levmax++;
                    else small *= 0.5;
                    erlarg = errsum;
                } while (0);
                if (busy) {
                    last++;
                    if (last > limit) busy = false;
                }
                DT_ADD(7, t_book);   // everything after the rules of this bisection (includes 4, 5 and 11)
            }
            // ---- the next bisection: the interval with the largest error estimate ------------------------------------------
            DT_MARK(t_top);
            if (busy && last > ATTD_K) {   // the list would outgrow its LDS column: the group's ray goes to the general kernel
                ovf = true;
                busy = false;
            }
            if ((__ballot(ovf) & gmask) != 0ULL) { ovf = true; busy = false; }   // an overflow of one lane ends its whole group
            if (__ballot(busy) == 0ULL) break;
            if (busy) {
                tn = node_of(maxerr);
                if (qagp) levcur = (int)L.tdepth[gi][tn] + 1;
                erlast = errmax;
            }
            pend = busy;
            two = true;
            DT_ADD(3, t_top);
        }
        DT_MARK(t_fin);
        nxt = load_rays(pair + pair_stride);   // in flight while this pair's results are finished and stored
        if (valid && entered_loop && !ovf) {
            if (last > limit) last = limit;
            bool sum_list = (exit_code == 1);
            if (!sum_list) {
                if (abserr == oflow) sum_list = true;
                else {
                    bool to_div_test = true;
                    if (ier + ierro != 0) {
                        if (ierro == 3) abserr += correc;
                        if (ier == 0) ier = 3;
                        if (result != 0. && area != 0.) {
                            if (abserr / fabs(result) > errsum / fabs(area)) { sum_list = true; to_div_test = false; }
                        } else {
                            if (abserr > errsum) { sum_list = true; to_div_test = false; }
                            else if (area == 0.) to_div_test = false;
                        }
                    }
                    if (to_div_test) {
                        if (!(ksgn == -1 && fmax(fabs(result), fabs(area)) <= resabs * 0.01)) {
                            if (0.01 > (result / area) || (result / area) > 100. || errsum > fabs(area)) ier = 6;
                        }
                    }
                }
            }
            if (sum_list) {
                result = 0.;
                for (int k = 1; k <= last; k++) result += L.rlist[k - 1][lane];
                abserr = errsum;
            }
            if (!qagp) neval_ = 42 * last - 21;
        }
        if (qagp && gz1 > gz2m) result *= -1.;
        if (ray_ok && !ovf) {
            const long item = gray * n_freq + jf;
            att[item] = valid ? det_exp(-1 * result) : NAN;
            if (neval) neval[item] = valid ? neval_ : 0;
            my_evals += (unsigned long long)(valid ? neval_ : 0);
        }
        if (g >= 0 && lane == glane0 && gray >= 0 && ovf) {
            const int slot = atomicAdd(overflow_count, 1);
            overflow_list[slot] = (int)gray;
        }
        wave_sync();   // the next pair's roots overwrite the tree
        DT_ADD(8, t_fin);
    }
#ifdef NRHIP_ATT_TIMING
    if (lane == 0) {
        for (int i = 0; i < 12; i++)
            if (i != 10) atomicAdd(&g_att_clk[i], tacc[i]);
        atomicAdd(&g_att_clk[10], __builtin_amdgcn_s_memtime() - t_kernel);
    }
#endif
    if (eval_counter) {
        for (int off = 32; off > 0; off >>= 1) my_evals += __shfl_xor(my_evals, off);
        if (lane == 0 && my_evals) atomicAdd(eval_counter, my_evals);
    }
}
