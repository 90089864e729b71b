"""CPU restatement of the No-U-Turn sampler of glmmrmcml_amd/csrc/nuts.h -- TEST INFRASTRUCTURE ONLY.

What it stands for: the reference samples the random effects with Stan through cmdstanr when usestan = TRUE
(R/gen_u_samples.R:38-69, R6ModelExtMCML.R:234-257; programs inst/stan/mcml_{gaussian,binomial,poisson}.stan:
gamma ~ std_normal(), y ~ family(Xb + Z gamma) with Z := Z L).  cmdstan is a third-party dependency that does not exist
in this image and the reference holds no captured Stan output: PARITY UNPINNED.  This file restates the published
algorithm (Stan's base_nuts: multinomial NUTS, Betancourt 2017, "A Conceptual Introduction to Hamiltonian Monte Carlo",
appendix A; Hoffman & Gelman 2014 for the dual averaging) in the same iterative form and with the same random streams
as the device code, one chain at a time in plain Python, so that tree depths, leapfrog counts, divergences and the
accepted subtrees can be compared transition by transition.  Log density and gradient: oracle.log_prob / log_grad
(mcmlmodel.h:138-153, 156-279).

U-turn checks: around every merged subtree and, as base_nuts.hpp does since Stan 2.23, between its two halves; the same
three checks for the whole tree after every doubling.
Metric: diag_e with Stan's windowed adaptation (windowed_adaptation.hpp, var_adaptation.hpp, welford_var_estimator.hpp,
adapt_diag_e_nuts.hpp), or unit_e.
Streams (shared with csrc/nuts.h): initial state 4 u - 2 with u = rng_uniform(seed, q, chain, 0, 16 iter_idx + 6); momentum of transition
`it` tag 16 iter_idx + 4 (divided by sqrt of the inverse metric); momentum of round r of the k-th step-size search tag
16 iter_idx + 5 with prop = 1000 k + r; uniforms from the
chain's minstd stream in the order: direction of a doubling; one draw per merge of a leaf (levels ascending); one
draw for the tree-level acceptance only when the subtree is not heavier than the tree.
"""
import ctypes as C
import math

import numpy as np

from . import oracle as orc


def _logaddexp(a, b):
    if a == -math.inf:
        return b
    if a == math.inf and b == math.inf:
        return math.inf
    if a > b:
        return a + math.log1p(math.exp(b - a))
    return b + math.log1p(math.exp(a - b))


class _Stream:
    def __init__(self, seed, chain, iter_idx):
        L = orc.lib()
        L.orc_chain_minstd_seed.restype = C.c_uint32
        L.orc_chain_minstd_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_minstd_canonical.restype = C.c_double
        self.x = C.c_uint32(L.orc_chain_minstd_seed(seed, chain, iter_idx))
        self.L = L

    def u(self):
        return self.L.orc_minstd_canonical(C.byref(self.x))


class _Windows:
    """stan/mcmc/windowed_adaptation.hpp"""

    def __init__(self, W):
        self.W, self.init, self.term, self.base, self.counter = W, 75, 50, 25, 0
        if W >= 20 and self.init + self.base + self.term > W:
            self.init = int(0.15 * W); self.term = int(0.1 * W); self.base = W - (self.init + self.term)
        self.next = self.init + self.base - 1
        self.size = self.base

    def in_window(self):
        return self.counter >= self.init and self.counter < self.W - self.term and self.counter != self.W

    def at_end(self):
        return self.counter == self.next and self.counter != self.W

    def compute_next(self):
        if self.next == self.W - self.term - 1:
            return
        self.size *= 2
        self.next = self.counter + self.size
        if self.next == self.W - self.term - 1:
            return
        if self.next + 2 * self.size >= self.W - self.term:
            self.next = self.W - self.term - 1


def nuts_chain(xb, ZL, y, var_par, fl, warmup, ndraw, seed, chain_id=0, iter_idx=0, max_treedepth=10, adapt_delta=0.8,
               stepsize=1.0, metric="diag_e"):
    """one chain: returns (draws Q x ndraw of gamma, trace dict of per-transition depth / nleap / eps / accept,
    diag dict)"""
    xb = np.ascontiguousarray(xb, dtype=float); y = np.ascontiguousarray(y, dtype=float)
    ZL = np.asfortranarray(ZL, dtype=float)
    Q = ZL.shape[1]
    lp_f = lambda v: orc.log_prob(xb, ZL, y, var_par, fl, v)
    gr_f = lambda v: np.asarray(orc.log_grad(xb, ZL, y, var_par, fl, v), dtype=float)
    normal = lambda prop, tag: np.array([orc.normal(seed, q, chain_id, prop, tag) for q in range(Q)])
    gen = _Stream(seed, chain_id, iter_idx)

    def uniform(prop, tag):                                       # csrc/rng.h::rng_uniform
        out = np.empty(Q)
        for q in range(Q):
            o = orc.philox((q, chain_id, prop, tag), (seed & 0xffffffff, (seed >> 32) & 0xffffffff))
            out[q] = ((o[0] >> 6) * 67108864.0 + (o[1] >> 6) + 0.5) * (1.0 / 4503599627370496.0)
        return out

    theta = 4.0 * uniform(0, 16 * iter_idx + 6) - 2.0              # Stan's default initial values: uniform(-2, 2)
    eps = float(stepsize)
    mi = np.ones(Q)                                               # inverse metric (diag_e starts from ones)

    def leapfrog(th, r, g, es):
        rh = r + (0.5 * es) * g
        x = th + es * (mi * rh)
        gn = gr_f(x)
        rn = rh + (0.5 * es) * gn
        return x, rn, gn

    def kinetic2(r):
        return float(np.dot(r, mi * r))

    def energy(x, r):
        h = -1 * lp_f(x) + 0.5 * kinetic2(r)
        return math.inf if math.isnan(h) else h

    # ---- init_stepsize (base_hmc.hpp)
    thr = math.log(0.8)
    state = dict(search_leaps=0, nsearch=0)

    def find_stepsize(eps):
        base = 1000 * state["nsearch"]
        state["nsearch"] += 1
        direction = 0
        for rnd in range(80):
            r0 = normal(base + rnd, 16 * iter_idx + 5) / np.sqrt(mi)
            H0 = -1 * lp_f(theta) + 0.5 * kinetic2(r0)
            x, rn, _ = leapfrog(theta, r0, gr_f(theta), eps)
            state["search_leaps"] += 1
            dH = H0 - energy(x, rn)
            if rnd == 0:
                direction = 1 if dH > thr else -1
                continue
            if (direction == 1 and not dH > thr) or (direction == -1 and not dH < thr):
                break
            e2 = 2 * eps if direction == 1 else 0.5 * eps
            if e2 > 1e7 or e2 < 1e-300:
                break
            eps = e2
        return eps

    eps = find_stepsize(eps)
    mu = math.log(10 * eps)
    counter, sbar, xbar = 0, 0.0, 0.0
    win = _Windows(warmup)
    wn, wm, ws = 0, np.zeros(Q), np.zeros(Q)

    total = warmup + ndraw
    draws = np.zeros((Q, ndraw), order="F")
    tr = dict(depth=np.zeros(total, dtype=int), nleap=np.zeros(total, dtype=int), eps=np.zeros(total),
              accept=np.zeros(total), ndiv=0, nhit=0)
    for it in range(total):
        r0 = normal(it, 16 * iter_idx + 4) / np.sqrt(mi)
        g0 = gr_f(theta)
        H0 = -1 * lp_f(theta) + 0.5 * kinetic2(r0)
        tm, rm, gm = theta.copy(), r0.copy(), g0.copy()          # backward edge
        tp, rp, gp = theta.copy(), r0.copy(), g0.copy()          # forward edge
        t_rho, t_th, lw_tree = r0.copy(), theta.copy(), 0.0
        depth, nleap, sum_acc = 0, 0, 0.0
        active = True
        for j in range(max_treedepth):
            d = 1 if gen.u() > 0.5 else -1
            padj = (rp if d > 0 else rm).copy()                   # the edge this doubling grows from, before it grows
            stack = {}                                            # level -> (rho, p_first, p_last, proposal, lw)
            valid = True
            for n in range(1 << j):
                tz = 0
                while (n >> tz) & 1:
                    tz += 1
                if d > 0:
                    tp, rp, gp = leapfrog(tp, rp, gp, d * eps); x, rn = tp, rp
                else:
                    tm, rm, gm = leapfrog(tm, rm, gm, d * eps); x, rn = tm, rm
                h = energy(x, rn)
                nleap += 1
                if (h - H0) > 1000.0:
                    tr["ndiv"] += 1
                    valid = False
                    break
                sum_acc += 1.0 if H0 - h > 0 else math.exp(H0 - h)
                lw = H0 - h
                choose = []
                for l in range(tz):
                    lwm = _logaddexp(stack[l][4], lw)
                    choose.append(gen.u() < math.exp(lw - lwm))
                    lw = lwm
                rho, pb, pe, th = rn.copy(), rn.copy(), rn.copy(), x.copy()   # a leaf: first- and last-built momentum
                for l in range(tz):
                    l_rho, l_pb, l_pe, l_th, _ = stack[l]
                    r_rho, r_pb, r_pe = rho, pb, pe
                    rho = l_rho + r_rho
                    pb = l_pb
                    if not choose[l]:
                        th = l_th
                    e1 = l_rho + r_pb                              # between the subtrees (base_nuts.hpp since 2.23)
                    e2 = r_rho + l_pe
                    dots = (np.dot(mi * l_pb, rho), np.dot(mi * r_pe, rho), np.dot(mi * l_pb, e1), np.dot(mi * r_pb, e1),
                            np.dot(mi * l_pe, e2), np.dot(mi * r_pe, e2))
                    if not all(float(v) > 0 for v in dots):
                        valid = False
                        break
                if not valid:
                    break
                stack[tz] = (rho, pb, pe, th, lw)
            if not valid:
                active = False
                break
            depth = j + 1
            s_rho, s_pb, _, s_th, lws = stack[j]
            if lws > lw_tree:
                acc = True
            else:
                acc = gen.u() < math.exp(lws - lw_tree)
            lw_tree = _logaddexp(lw_tree, lws)
            rho_old = t_rho
            t_rho = rho_old + s_rho
            if acc:
                t_th = s_th
            far_old, new_edge = (rm, rp) if d > 0 else (rp, rm)
            e1 = rho_old + s_pb
            e2 = s_rho + padj
            dots = (np.dot(mi * rm, t_rho), np.dot(mi * rp, t_rho), np.dot(mi * far_old, e1), np.dot(mi * s_pb, e1),
                    np.dot(mi * padj, e2), np.dot(mi * new_edge, e2))
            if not all(float(v) > 0 for v in dots):
                active = False
                break
            if depth >= max_treedepth:
                tr["nhit"] += 1
                break
        theta = t_th.copy()
        stat = sum_acc / nleap if nleap > 0 else 0.0
        tr["depth"][it] = depth; tr["nleap"][it] = nleap; tr["eps"][it] = eps
        if it < warmup:
            counter += 1
            stat = min(stat, 1.0)
            eta = 1.0 / (counter + 10.0)
            sbar = (1.0 - eta) * sbar + eta * (adapt_delta - stat)
            xx = mu - sbar * math.sqrt(counter) / 0.05
            xeta = counter ** -0.75
            xbar = (1.0 - xeta) * xbar + xeta * xx
            eps = math.exp(xbar) if it == warmup - 1 else math.exp(xx)
            if metric == "diag_e":                                # var_adaptation::learn_variance
                if win.in_window():
                    wn += 1
                    delta = theta - wm
                    wm = wm + delta / wn
                    ws = ws + (theta - wm) * delta
                update = win.at_end()
                if update:
                    win.compute_next()
                    var = ws / (wn - 1.0)
                    mi = (wn / (wn + 5.0)) * var + 1e-3 * (5.0 / (wn + 5.0))
                    wn, wm, ws = 0, np.zeros(Q), np.zeros(Q)
                win.counter += 1
                if update:                                        # adapt_diag_e_nuts::transition
                    eps = find_stepsize(eps)
                    mu = math.log(10 * eps)
                    counter, sbar, xbar = 0, 0.0, 0.0
        tr["accept"][it] = stat
        if it >= warmup:
            draws[:, it - warmup] = theta
    return draws, tr, dict(eps=eps, search_leaps=state["search_leaps"], inv_metric=mi)
