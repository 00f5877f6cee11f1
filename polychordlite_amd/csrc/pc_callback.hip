// pc_callback.hip -- slice-sampling chains whose likelihood is a HOST callback.
//
// The drop-in use of the reference is a user loglikelihood(theta) -> (logL, phi) and prior(cube) ->
// theta in C / Fortran / Python.  Those cannot run inside a kernel, so the device only
// PROPOSES: k_slice_tick advances every chain's slice-sampling state machine (the same arithmetic and
// the same Philox draws as k_slice: chordal_sampling.f90:163-273) until it needs a likelihood value,
// parks the proposal's hypercube coordinates, and returns; the host evaluates prior + likelihood for
// the parked proposals on the calling thread (the reference's threading contract,
// _pypolychord.cpp:219) and relaunches.  Out-of-cube proposals never reach the host
// (calculate.f90:36-38).  Everything downstream (contraction, phantoms, covariance, clustering) is
// unchanged.
#include "pc_state.h"

enum { CB_NEW_SLICE = 0, CB_WAIT_R0, CB_WAIT_L0, CB_STEP_R, CB_WAIT_R, CB_STEP_L, CB_WAIT_L, CB_SHRINK, CB_WAIT_S, CB_DONE };

struct PcChain {
    int phase, s, istep, it, kdraw, nlike, need, ok_theta;
    double t, tL, tR, lL, lR, w, contour, lnew;
};

// One wavefront per chain, lane = cube coordinate (d, d + 64, ...): a proposal, an acceptance or a row store touches all
// coordinates in one round trip (one thread per chain walked them one global access at a time: 68 us per tick at 20-D).
// Everything that steers the state machine is wave-uniform.
__global__ __launch_bounds__(64) void k_slice_tick(PcState S, unsigned batch, int nchains, PcChain *cs, double *x0s /* [B][D] */,
                                                  int *decks /* [B][nr] */, double *prop /* [B][D] */,
                                                  const double *ev_logL, const double *ev_theta, const double *ev_phi,
                                                  int first, double *prop_host, int *need_host)
{   // prop_host / need_host: pinned host memory; the proposals and the need flags are stored there directly (posted
    // writes over PCIe), so that the host finds them after one stream synchronisation, without copies
    const int chain = blockIdx.x, lane = threadIdx.x;
    if (chain >= nchains) return;
    const int D = S.D, nr = S.nr, nT = S.nT;
    PcChain c = cs[chain];
    double *x0 = x0s + (size_t)chain * D;
    int *deck = decks + (size_t)chain * nr;
    double *pc = prop + (size_t)chain * D;
    if (first) {
        const double *seed = S.live + (size_t)S.ch_seed_slot[chain] * nT;
        for (int d = lane; d < D; d += 64) x0[d] = seed[d];
        for (int i = lane; i < nr; i += 64) deck[i] = i;
        __syncthreads();
        if (lane == 0)
            for (int i = nr - 1; i >= 1; --i) {            // random_utils.F90:505-532 on deck(2:)
                const double u = pc_uniform(S.k0, S.k1, PC_DOM_SHUFFLE, batch, (uint32_t)chain, (uint32_t)i);
                int j = (int)ceil(u * i);
                j = j < 1 ? 1 : (j > i ? i : j);
                const int t = deck[i]; deck[i] = deck[j]; deck[j] = t;
            }
        __syncthreads();
        c.phase = CB_NEW_SLICE; c.s = 0; c.nlike = 0; c.need = 0; c.contour = S.ch_contour[chain]; c.ok_theta = 0;
        if (S.ngrade > 1 && lane < PC_MAX_GRADE) S.ch_nlike_g[(size_t)chain * PC_MAX_GRADE + lane] = 0;
        __syncthreads();
    }
    double logL = 0.0;
    bool have = false;
    if (c.need) {                                        // the host answered the parked proposal
        logL = ev_logL[chain];
        if (logL > S.logzero) {
            c.nlike++;
            if (S.ngrade > 1 && lane == 0) S.ch_nlike_g[(size_t)chain * PC_MAX_GRADE + pc_grade_of(S, deck[c.s])]++;   // chordal_sampling.f90:84
        }
        c.need = 0; have = true; c.ok_theta = 1;
    }
    const double *nh = S.nhat + ((size_t)chain * nr + deck[c.s < nr ? c.s : 0]) * D;
    auto next_u = [&]() { const uint32_t k = (uint32_t)c.kdraw++; return pc_uniform(S.k0, S.k1, PC_DOM_SLICE, batch, (uint32_t)chain, (uint32_t)c.s * PC_SLICE_STRIDE + k); };
    // park a proposal; returns true if the host must evaluate it, false if it is outside the cube
    auto propose = [&](double t) {
        c.t = t;
        bool outside = false;
        double *ph = prop_host + (size_t)chain * D;
        for (int d = lane; d < D; d += 64) { const double v = x0[d] + t * nh[d]; pc[d] = v; ph[d] = v; outside |= (v < 0.0) | (v > 1.0); }
        if (__ballot(outside) != 0ull) { logL = S.logzero; have = true; c.ok_theta = 0; return false; }   // calculate.f90:36-38
        c.need = 1; have = false;
        return true;
    };
    // every iteration either takes a decision (and possibly parks a proposal) or consumes an answer
    for (int guard = 0; guard < 1000000; ++guard) {
        bool parked = false, accept = false;
        switch (c.phase) {
        case CB_NEW_SLICE:
            if (c.s >= nr) { c.phase = CB_DONE; break; }
            nh = S.nhat + ((size_t)chain * nr + deck[c.s]) * D;
            c.w = S.nhat_w[(size_t)chain * nr + deck[c.s]];
            c.kdraw = 0;
            {   // initial bracket (chordal_sampling.f90:213-219)
                const double u0 = next_u();
                c.tR = (1 - u0) * c.w; c.tL = -(u0 * c.w);
            }
            c.phase = CB_WAIT_R0; parked = propose(c.tR);
            break;
        case CB_WAIT_R0:
            if (!have) { parked = true; break; }
            have = false; c.lR = logL; c.phase = CB_WAIT_L0; parked = propose(c.tL);
            break;
        case CB_WAIT_L0:
            if (!have) { parked = true; break; }
            have = false; c.lL = logL; c.istep = 0; c.phase = CB_STEP_R;
            break;
        case CB_STEP_R:   // stepping out (:223-227)
            if (c.lR >= c.contour && c.lR > S.logzero) { c.istep++; c.tR = c.w * c.istep; c.phase = CB_WAIT_R; parked = propose(c.tR); }
            else { c.istep = 0; c.phase = CB_STEP_L; }
            break;
        case CB_WAIT_R:
            if (!have) { parked = true; break; }
            have = false; c.lR = logL; c.phase = CB_STEP_R;
            break;
        case CB_STEP_L:   // (:232-236)
            if (c.lL >= c.contour && c.lL > S.logzero) { c.istep++; c.tL = -(c.w * c.istep); c.phase = CB_WAIT_L; parked = propose(c.tL); }
            else { c.it = 0; c.phase = CB_SHRINK; }
            break;
        case CB_WAIT_L:
            if (!have) { parked = true; break; }
            have = false; c.lL = logL; c.phase = CB_STEP_L;
            break;
        case CB_SHRINK:   // shrinkage (:240-271), at most 101 trials
            if (c.it > 100) { c.lnew = S.logzero; accept = true; }
            else {
                const double dl = fabs(c.tL), dr = fabs(c.tR);
                const double t = next_u() * (dr + dl) - dl;
                c.phase = CB_WAIT_S; parked = propose(t);
            }
            break;
        case CB_WAIT_S:
            if (!have) { parked = true; break; }
            have = false; c.lnew = logL;
            if (c.lnew < c.contour || c.lnew <= S.logzero) { if (c.t > 0.0) c.tR = c.t; else c.tL = c.t; c.it++; c.phase = CB_SHRINK; }
            else accept = true;
            break;
        default: break;
        }
        if (accept) {   // the baby becomes the next start point (chordal_sampling.f90:85-88)
            double *row = S.babies + ((size_t)chain * nr + c.s) * nT;
            for (int d = lane; d < D; d += 64) {
                const double v = pc[d];
                x0[d] = v; row[d] = v; row[S.p0 + d] = c.ok_theta ? ev_theta[(size_t)chain * D + d] : 0.0;
            }
            for (int e = lane; e < S.nDer; e += 64) row[S.d0 + e] = c.ok_theta ? ev_phi[(size_t)chain * S.nDer + e] : 0.0;
            if (lane == 0) {
                row[S.b0] = c.contour; row[S.l0] = c.lnew;
                S.baby_logL[(size_t)chain * nr + c.s] = c.lnew; S.baby_logL_T[(size_t)c.s * S.B + chain] = c.lnew;
            }
            __syncthreads();                       // x0 of the next slice is read by other lanes than wrote it?  no: lane-private
            c.s++; c.phase = CB_NEW_SLICE;
        }
        if (c.phase == CB_DONE) break;
        if (parked && c.need) break;       // a proposal waits for the host
    }
    if (lane == 0) {
        if (c.phase == CB_DONE) S.ch_nlike[chain] = c.nlike;
        need_host[chain] = c.need;
        cs[chain] = c;
    }
}

extern "C" void pc_launch_slice_tick(const PcState *S, unsigned batch, int nchains, void *cs, double *x0s, int *decks, double *prop,
                                     const double *ev_logL, const double *ev_theta, const double *ev_phi, int first, double *prop_host,
                                     int *need_host, hipStream_t st)
{
    hipLaunchKernelGGL(k_slice_tick, dim3(nchains), dim3(64), 0, st, *S, batch, nchains, (PcChain *)cs, x0s, decks, prop,
                       ev_logL, ev_theta, ev_phi, first, prop_host, need_host);
}

extern "C" size_t pc_chain_state_size(void) { return sizeof(PcChain); }
