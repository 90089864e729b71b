// optim.hip -- host code only (see optim.h): BOBYQA-family optimiser and R-style
// finite-difference gradient / Hessian.
#include "optim.h"
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstdio>

namespace mcml {
namespace {

typedef std::vector<double> vec;

static double dot(const vec& a, const vec& b) { double s = 0; for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i]; return s; }
static double norm2(const vec& a) { return std::sqrt(dot(a, a)); }

// in-place inverse by Gauss-Jordan with partial pivoting; false if singular
static bool invert(std::vector<double>& A, int N)
{
    std::vector<double> B((size_t)N * N, 0.0);
    for (int i = 0; i < N; ++i) B[i + (size_t)i * N] = 1.0;
    double amax = 0;
    for (double v : A) amax = std::max(amax, std::fabs(v));
    for (int c = 0; c < N; ++c) {
        int p = c; double best = std::fabs(A[c + (size_t)c * N]);
        for (int i = c + 1; i < N; ++i) if (std::fabs(A[i + (size_t)c * N]) > best) { best = std::fabs(A[i + (size_t)c * N]); p = i; }
        if (!(best > 1e-14 * amax)) return false;
        if (p != c) for (int j = 0; j < N; ++j) { std::swap(A[c + (size_t)j * N], A[p + (size_t)j * N]); std::swap(B[c + (size_t)j * N], B[p + (size_t)j * N]); }
        double d = A[c + (size_t)c * N];
        for (int j = 0; j < N; ++j) { A[c + (size_t)j * N] /= d; B[c + (size_t)j * N] /= d; }
        for (int i = 0; i < N; ++i) if (i != c) {
            double f = A[i + (size_t)c * N];
            if (f != 0.0) for (int j = 0; j < N; ++j) { A[i + (size_t)j * N] -= f * A[c + (size_t)j * N]; B[i + (size_t)j * N] -= f * B[c + (size_t)j * N]; }
        }
    }
    A.swap(B);
    return true;
}

struct Quad {          // q(x) = c + g'(x - xb) + 1/2 (x - xb)' H (x - xb)
    double c = 0; vec g, H; int n = 0;
    void init(int n_) { n = n_; c = 0; g.assign(n, 0.0); H.assign((size_t)n * n, 0.0); }
    vec Hv(const vec& d) const { vec r(n, 0.0); for (int j = 0; j < n; ++j) { double dj = d[j]; if (dj != 0.0) for (int i = 0; i < n; ++i) r[i] += H[i + (size_t)j * n] * dj; } return r; }
    double eval(const vec& d) const { vec h = Hv(d); return c + dot(g, d) + 0.5 * dot(d, h); }
    void shift(const vec& d) {      // move the base point by d
        vec h = Hv(d);
        c = c + dot(g, d) + 0.5 * dot(d, h);
        for (int i = 0; i < n; ++i) g[i] += h[i];
    }
};

// minimise g'd + 1/2 d'Hd  s.t. ||d|| <= delta, a <= d <= b (a <= 0 <= b):
// truncated conjugate gradients on the free variables, variables that hit a
// bound are fixed and CG restarts (the role of Powell's TRSBOX).
static vec trust_step(const Quad& q, double delta, const vec& a, const vec& b, double* crvmin)
{
    const int n = q.n;
    vec d(n, 0.0);
    std::vector<char> fixed(n, 0);
    for (int i = 0; i < n; ++i)
        if ((a[i] >= 0.0 && q.g[i] >= 0.0) || (b[i] <= 0.0 && q.g[i] <= 0.0)) fixed[i] = 1;
    double cmin = -1.0;
    const double g0 = norm2(q.g);
    for (int restart = 0; restart <= n; ++restart) {
        vec h = q.Hv(d), r(n), p(n);
        for (int i = 0; i < n; ++i) r[i] = fixed[i] ? 0.0 : -(q.g[i] + h[i]);
        p = r;
        double rr = dot(r, r);
        bool hit_bound = false;
        for (int it = 0; it < n && rr > 1e-24 * std::max(1.0, g0 * g0); ++it) {
            vec hp = q.Hv(p);
            for (int i = 0; i < n; ++i) if (fixed[i]) hp[i] = 0.0;
            const double php = dot(p, hp), pp = dot(p, p);
            if (pp <= 0.0) break;
            if (php > 0.0) { double cv = php / pp; cmin = (cmin < 0.0) ? cv : std::min(cmin, cv); } else cmin = 0.0;
            double alpha = (php > 0.0) ? rr / php : HUGE_VAL;
            // ball
            const double dp = dot(d, p), dd = dot(d, d);
            double disc = dp * dp + pp * (delta * delta - dd);
            double aball = (-dp + std::sqrt(std::max(0.0, disc))) / pp;
            // bounds
            double abnd = HUGE_VAL; int ib = -1;
            for (int i = 0; i < n; ++i) if (!fixed[i] && p[i] != 0.0) {
                double t = p[i] > 0.0 ? (b[i] - d[i]) / p[i] : (a[i] - d[i]) / p[i];
                if (t < abnd) { abnd = t; ib = i; }
            }
            double step = std::min(alpha, std::min(aball, abnd));
            if (step < 0.0) step = 0.0;
            for (int i = 0; i < n; ++i) d[i] += step * p[i];
            if (step == abnd && abnd <= aball && abnd <= alpha && ib >= 0) {
                d[ib] = p[ib] > 0.0 ? b[ib] : a[ib];
                fixed[ib] = 1; hit_bound = true; break;
            }
            if (step == aball && aball <= alpha) { if (crvmin) *crvmin = 0.0; return d; }
            for (int i = 0; i < n; ++i) r[i] -= step * hp[i];
            const double rr2 = dot(r, r);
            const double beta = rr2 / rr;
            rr = rr2;
            for (int i = 0; i < n; ++i) p[i] = r[i] + beta * p[i];
        }
        if (!hit_bound) break;
    }
    if (crvmin) *crvmin = cmin < 0.0 ? 0.0 : cmin;
    return d;
}

class Bobyqa {
public:
    Bobyqa(const objective_fn& f, const vec& x0, const vec& lo, const vec& up, const BobyqaOpts& o)
        : f_(f), n_((int)x0.size()), lo_(lo), up_(up), o_(o), x0_(x0) {}
    Bobyqa(const batch_objective_fn& fb, const vec& x0, const vec& lo, const vec& up, const BobyqaOpts& o)
        : f_(none_), fb_(&fb), n_((int)x0.size()), lo_(lo), up_(up), o_(o), x0_(x0) {}

    int run(BobyqaResult* res)
    {
        const int n = n_;
        npt_ = o_.npt > 0 ? o_.npt : std::min(n + 2, 2 * n);
        npt_ = std::max(npt_, std::min(n + 2, (n + 1) * (n + 2) / 2));
        npt_ = std::min(npt_, 2 * n + 1);
        double xmax = 0; for (double v : x0_) xmax = std::max(xmax, std::fabs(v));
        double rhobeg = o_.rhobeg > 0 ? o_.rhobeg : std::min(0.95, 0.2 * xmax);
        if (!(rhobeg > 0)) rhobeg = 0.1;
        for (int i = 0; i < n; ++i) {
            double rng = up_[i] - lo_[i];
            if (std::isfinite(rng)) { if (!(rng > 0)) { set_error("bobyqa: empty bound interval"); return MCML_EINVAL; } rhobeg = std::min(rhobeg, 0.5 * rng * 0.999); }
        }
        // BOBYQA moves a start that is closer than rhobeg to a bound onto that bound.  For the
        // covariance parameters (lower bound 1e-6, mcmloptim.h:35-38) that means evaluating the
        // MVN likelihood at a variance of 1e-12: an objective ~1e11 that wrecks the first models.
        // When no rhobeg was asked for, keep an interior start interior instead: the default is
        // also capped by half the distance to the nearest bound.  A coordinate that sits within
        // 1e-3 of the nominal radius of a bound (e.g. a variance an earlier fit drove onto 1e-6, give
        // or take round-off) IS on that bound: it is snapped there instead of shrinking the radius
        // to nothing.
        if (!(o_.rhobeg > 0)) {
            const double near = 1e-3 * rhobeg;
            for (int i = 0; i < n; ++i) {
                double xi = std::min(std::max(x0_[i], lo_[i]), up_[i]);
                const double dl = xi - lo_[i], du = up_[i] - xi;
                if (std::isfinite(dl) && dl > 0) { if (dl < near) x0_[i] = lo_[i]; else rhobeg = std::min(rhobeg, 0.5 * dl); }
                if (std::isfinite(du) && du > 0) { if (du < near) x0_[i] = up_[i]; else rhobeg = std::min(rhobeg, 0.5 * du); }
            }
        }
        const double rhoend = o_.rhoend > 0 ? std::min(o_.rhoend, rhobeg) : 1e-6 * rhobeg;
        // x0 is moved so that every coordinate is either on a bound or >= rhobeg from it (BOBYQA)
        vec x = x0_;
        for (int i = 0; i < n; ++i) {
            x[i] = std::min(std::max(x[i], lo_[i]), up_[i]);
            if (x[i] - lo_[i] < rhobeg && x[i] != lo_[i]) x[i] = (x[i] - lo_[i] < 0.5 * rhobeg) ? lo_[i] : lo_[i] + rhobeg;
            if (up_[i] - x[i] < rhobeg && x[i] != up_[i]) x[i] = (up_[i] - x[i] < 0.5 * rhobeg) ? up_[i] : up_[i] - rhobeg;
        }
        nf_ = 0; rc_ = 0;
        Y_.assign(npt_, x); F_.assign(npt_, 0.0);
        // initial points (PRELIM): x0, x0 +- rhobeg e_i
        for (int k = 1; k < npt_; ++k) {
            int i = (k - 1) % n; bool second = (k - 1) >= n;
            double step = rhobeg;
            if (!second) { if (up_[i] - x[i] < rhobeg * 0.999) step = -rhobeg; }
            else { step = -rhobeg; if (x[i] - lo_[i] < rhobeg * 0.999) step = 2 * rhobeg; if (up_[i] - x[i] < rhobeg * 0.999) step = -2 * rhobeg; }
            Y_[k][i] = std::min(std::max(x[i] + step, lo_[i]), up_[i]);
        }
        for (int k = 0; k < npt_; ++k) { F_[k] = eval(Y_[k]); if (rc_) return rc_; }
        kopt_ = (int)(std::min_element(F_.begin(), F_.end()) - F_.begin());
        q_.init(n); q_.c = 0; xb_ = Y_[kopt_];
        if (!refit(true)) { set_error("bobyqa: initial interpolation set is degenerate"); return MCML_EINVAL; }

        double rho = rhobeg, delta = rho;
        int ntrits = 0, nfsav = nf_;
        double diffa = 0, diffb = 0, diffc = 0, ratio = 0, dnorm = 0;
        int status = 0;
        enum { TRSTEP, GEOM, REDUCE, DONE } state = TRSTEP;
        double distsq = 0;
        vec d(n, 0.0);
        while (state != DONE) {
            if (nf_ >= o_.maxfun) { status = 1; break; }
            if (state == TRSTEP) {
                move_base_to_opt();
                vec a(n), b(n);
                for (int i = 0; i < n; ++i) { a[i] = lo_[i] - xb_[i]; b[i] = up_[i] - xb_[i]; }
                double crvmin = 0;
                d = trust_step(q_, delta, a, b, &crvmin);
                dnorm = std::min(delta, norm2(d));
                if (o_.iprint > 1) fprintf(stderr, "bobyqa tr: dnorm=%.3g delta=%.3g rho=%.3g crvmin=%.3g |g|=%.3g\n", dnorm, delta, rho, crvmin, norm2(q_.g));
                if (dnorm < 0.5 * rho) {
                    ntrits = -1;
                    distsq = 100.0 * rho * rho;
                    if (nf_ <= nfsav + 2) { state = GEOM; continue; }
                    const double errbig = std::max(diffa, std::max(diffb, diffc));
                    const double frhosq = 0.125 * rho * rho;
                    if (crvmin > 0.0 && errbig > frhosq * crvmin) { state = GEOM; continue; }
                    state = REDUCE; continue;
                }
                ++ntrits;
                // evaluate the trial point
                vec xnew(n);
                for (int i = 0; i < n; ++i) xnew[i] = std::min(std::max(xb_[i] + d[i], lo_[i]), up_[i]);
                for (int i = 0; i < n; ++i) d[i] = xnew[i] - xb_[i];
                const double fopt = F_[kopt_];
                const double vquad = q_.eval(d) - q_.c;
                const double fnew = eval(xnew); if (rc_) return rc_;
                const double diff = fnew - fopt - vquad;
                diffc = diffb; diffb = diffa; diffa = std::fabs(diff);
                if (dnorm > rho) nfsav = nf_;
                if (!(vquad < 0.0)) { ratio = -1.0; }
                else ratio = (fnew - fopt) / vquad;
                const double hdelta = 0.5 * delta;
                if (ratio <= 0.1) delta = std::min(hdelta, dnorm);
                else if (ratio <= 0.7) delta = std::max(hdelta, dnorm);
                else delta = std::max(hdelta, 2.0 * dnorm);
                if (delta <= 1.5 * rho) delta = rho;
                int knew = pick_replace(xnew, fnew < fopt, delta);
                if (knew >= 0) { replace(knew, xnew, fnew); }
                if (o_.iprint > 1) fprintf(stderr, "bobyqa nf=%d f=%.10g rho=%.3g delta=%.3g ratio=%.3g\n", nf_, F_[kopt_], rho, delta, ratio);
                if (ratio >= 0.1) { state = TRSTEP; continue; }
                distsq = std::max(4.0 * delta * delta, 100.0 * rho * rho);
                state = GEOM; continue;
            }
            if (state == GEOM) {
                move_base_to_opt();
                int knew = -1; double dmax = distsq;
                for (int k = 0; k < npt_; ++k) {
                    double s = 0; for (int i = 0; i < n; ++i) { double t = Y_[k][i] - xb_[i]; s += t * t; }
                    if (s > dmax) { dmax = s; knew = k; }
                }
                if (knew >= 0) {
                    const double dist = std::sqrt(dmax);
                    if (o_.iprint > 1) fprintf(stderr, "bobyqa geom: replace %d dist=%.3g ntrits=%d delta=%.3g rho=%.3g\n", knew, dist, ntrits, delta, rho);
                    if (ntrits == -1) { delta = std::min(0.1 * delta, 0.5 * dist); if (delta <= 1.5 * rho) delta = rho; }
                    ntrits = 0;
                    const double adelt = std::max(std::min(0.1 * dist, delta), rho);
                    vec xnew;
                    if (!geometry_point(knew, adelt, &xnew)) { state = REDUCE; continue; }
                    const double fopt = F_[kopt_];
                    vec dd(n); for (int i = 0; i < n; ++i) dd[i] = xnew[i] - xb_[i];
                    const double vquad = q_.eval(dd) - q_.c;
                    const double fnew = eval(xnew); if (rc_) return rc_;
                    const double diff = fnew - fopt - vquad;
                    diffc = diffb; diffb = diffa; diffa = std::fabs(diff);
                    replace(knew, xnew, fnew);
                    state = TRSTEP; continue;
                }
                if (ntrits == -1) { state = REDUCE; continue; }
                if (ratio > 0.0) { state = TRSTEP; continue; }
                if (std::max(delta, dnorm) > rho) { state = TRSTEP; continue; }
                state = REDUCE; continue;
            }
            if (state == REDUCE) {
                if (rho > rhoend) {
                    delta = 0.5 * rho;
                    const double r = rho / rhoend;
                    if (r <= 16.0) rho = rhoend;
                    else if (r <= 250.0) rho = std::sqrt(r) * rhoend;
                    else rho = 0.1 * rho;
                    delta = std::max(delta, rho);
                    ntrits = 0; nfsav = nf_;
                    if (o_.iprint > 0) fprintf(stderr, "bobyqa: rho -> %.3g  nf=%d  f=%.12g\n", rho, nf_, F_[kopt_]);
                    state = TRSTEP; continue;
                }
                if (ntrits == -1) {                    // one last evaluation at the short step
                    vec xnew(n);
                    for (int i = 0; i < n; ++i) xnew[i] = std::min(std::max(xb_[i] + d[i], lo_[i]), up_[i]);
                    bool moved = false; for (int i = 0; i < n; ++i) if (xnew[i] != Y_[kopt_][i]) moved = true;
                    if (moved && nf_ < o_.maxfun) {
                        const double fnew = eval(xnew); if (rc_) return rc_;
                        if (fnew < F_[kopt_]) { Y_[kopt_] = xnew; F_[kopt_] = fnew; }
                    }
                }
                state = DONE;
            }
        }
        kopt_ = (int)(std::min_element(F_.begin(), F_.end()) - F_.begin());
        res->x = Y_[kopt_]; res->fval = F_[kopt_]; res->nfev = nf_; res->status = status;
        return MCML_OK;
    }

    // ---------------------------------------------------------------- the batch schedule (optim.h)
    int run_batch(int width, BobyqaResult* res)
    {
        const int n = n_;
        const int W = std::max(1, width);
        const int nfull = (n + 1) * (n + 2) / 2;
        if (o_.npt > 0) npt_ = std::min(std::max(o_.npt, n + 2), nfull);
        else npt_ = (nfull <= std::max(W, 2 * n + 1)) ? nfull : 2 * n + 1;
        // radius and start exactly as run() chooses them
        double xmax = 0; for (double v : x0_) xmax = std::max(xmax, std::fabs(v));
        double rhobeg = o_.rhobeg > 0 ? o_.rhobeg : std::min(0.95, 0.2 * xmax);
        if (!(rhobeg > 0)) rhobeg = 0.1;
        for (int i = 0; i < n; ++i) {
            double rng = up_[i] - lo_[i];
            if (std::isfinite(rng)) { if (!(rng > 0)) { set_error("bobyqa: empty bound interval"); return MCML_EINVAL; } rhobeg = std::min(rhobeg, 0.5 * rng * 0.999); }
        }
        if (!(o_.rhobeg > 0)) {
            const double near = 1e-3 * rhobeg;
            for (int i = 0; i < n; ++i) {
                double xi = std::min(std::max(x0_[i], lo_[i]), up_[i]);
                const double dl = xi - lo_[i], du = up_[i] - xi;
                if (std::isfinite(dl) && dl > 0) { if (dl < near) x0_[i] = lo_[i]; else rhobeg = std::min(rhobeg, 0.5 * dl); }
                if (std::isfinite(du) && du > 0) { if (du < near) x0_[i] = up_[i]; else rhobeg = std::min(rhobeg, 0.5 * du); }
            }
        }
        const double rhoend = o_.rhoend > 0 ? std::min(o_.rhoend, rhobeg) : 1e-6 * rhobeg;
        vec x = x0_;
        for (int i = 0; i < n; ++i) {
            x[i] = std::min(std::max(x[i], lo_[i]), up_[i]);
            if (x[i] - lo_[i] < rhobeg && x[i] != lo_[i]) x[i] = (x[i] - lo_[i] < 0.5 * rhobeg) ? lo_[i] : lo_[i] + rhobeg;
            if (up_[i] - x[i] < rhobeg && x[i] != up_[i]) x[i] = (up_[i] - x[i] < 0.5 * rhobeg) ? up_[i] : up_[i] - rhobeg;
        }
        nf_ = 0; rc_ = 0;
        int rounds = 0;
        auto clampv = [&](vec& y) { for (int i = 0; i < n; ++i) y[i] = std::min(std::max(y[i], lo_[i]), up_[i]); };
        auto dist = [&](const vec& p, const vec& q) { double t = 0; for (int i = 0; i < n; ++i) { double e = p[i] - q[i]; t += e * e; } return std::sqrt(t); };

        // ---- initial design: x, x +- rhobeg e_i (run()'s points), the pair points of a full quadratic; a last round
        // that is not full also takes the opposite diagonals (they compete for the best point only)
        std::vector<vec> P0(1, x);
        vec s1(n, 0.0);
        for (int i = 0; i < n; ++i) {
            double step = rhobeg;
            if (up_[i] - x[i] < rhobeg * 0.999) step = -rhobeg;
            vec y = x; y[i] = std::min(std::max(x[i] + step, lo_[i]), up_[i]); s1[i] = y[i] - x[i];
            P0.push_back(y);
        }
        for (int i = 0; i < n; ++i) {
            double step = -rhobeg;
            if (x[i] - lo_[i] < rhobeg * 0.999) step = 2 * rhobeg;
            if (up_[i] - x[i] < rhobeg * 0.999) step = -2 * rhobeg;
            vec y = x; y[i] = std::min(std::max(x[i] + step, lo_[i]), up_[i]);
            P0.push_back(y);
        }
        for (int i = 0; i < n && (int)P0.size() < npt_; ++i)
            for (int j = i + 1; j < n && (int)P0.size() < npt_; ++j) { vec y = x; y[i] += s1[i]; y[j] += s1[j]; P0.push_back(y); }
        P0.resize(npt_);
        const int budget0 = std::min(o_.maxfun, ((npt_ + W - 1) / W) * W);
        for (int i = 0; i < n && (int)P0.size() < budget0; ++i)
            for (int j = i + 1; j < n && (int)P0.size() < budget0; ++j) {
                vec y = x; y[i] += s1[i]; y[j] -= s1[j];
                if (y[j] < lo_[j] || y[j] > up_[j]) continue;
                P0.push_back(y);
            }
        if ((int)P0.size() > o_.maxfun) P0.resize(std::max(1, o_.maxfun));
        vec F0;
        for (size_t at = 0; at < P0.size(); at += W) {
            std::vector<vec> X(P0.begin() + at, P0.begin() + std::min(P0.size(), at + (size_t)W));
            vec F;
            if (!eval_batch(X, &F)) return rc_;
            ++rounds;
            F0.insert(F0.end(), F.begin(), F.end());
        }
        if ((int)P0.size() < npt_) {            // the budget ended inside the initial design
            size_t kb = std::min_element(F0.begin(), F0.end()) - F0.begin();
            res->x = P0[kb]; res->fval = F0[kb]; res->nfev = nf_; res->status = 1; res->rounds = rounds;
            return MCML_OK;
        }
        Y_.assign(P0.begin(), P0.begin() + npt_); F_.assign(F0.begin(), F0.begin() + npt_);
        kopt_ = (int)(std::min_element(F_.begin(), F_.end()) - F_.begin());
        q_.init(n); q_.c = 0; xb_ = Y_[kopt_];
        if (!refit(true)) { set_error("bobyqa: initial interpolation set is degenerate"); return MCML_EINVAL; }
        double rho = rhobeg, delta = rho;
        for (size_t k = npt_; k < P0.size(); ++k)
            if (F0[k] < F_[kopt_]) {
                move_base_to_opt();
                int knew = pick_replace(P0[k], true, delta);
                if (knew >= 0) replace(knew, P0[k], F0[k]);
                if (rc_) return rc_;
            }

        struct Cand { vec x; int kind; int knew; double radius, vquad, dnorm; };
        int status = 0;
        for (;;) {
            if (nf_ >= o_.maxfun) { status = 1; break; }
            move_base_to_opt();
            const int room = std::min(W, o_.maxfun - nf_);
            vec a(n), b(n);
            for (int i = 0; i < n; ++i) { a[i] = lo_[i] - xb_[i]; b[i] = up_[i] - xb_[i]; }
            vec d1 = trust_step(q_, delta, a, b, nullptr);
            const double dn1 = std::min(delta, norm2(d1));
            const bool short_step = dn1 < 0.5 * rho;
            // the rho that follows this one (Powell's schedule; after a short step never larger than twice that step)
            double rho_next = rho;
            {
                const double r = rho / rhoend;
                rho_next = (r <= 16.0) ? rhoend : (r <= 250.0) ? std::sqrt(r) * rhoend : 0.1 * rho;
                if (short_step) rho_next = std::max(rhoend, std::min(rho_next, std::max(2.0 * dn1, 1e-3 * rho)));
            }
            const bool last_level = !(rho > rhoend);
            if (o_.iprint > 1) fprintf(stderr, "bobyqa_batch: round %d nf=%d f=%.12g rho=%.3g delta=%.3g dnorm=%.3g%s\n", rounds, nf_, F_[kopt_], rho, delta, dn1, short_step ? " (short)" : "");
            std::vector<Cand> cs;
            auto push_tr = [&](const vec& d, double radius, bool dedup) {
                vec xn(n); for (int i = 0; i < n; ++i) xn[i] = std::min(std::max(xb_[i] + d[i], lo_[i]), up_[i]);
                vec dd(n); for (int i = 0; i < n; ++i) dd[i] = xn[i] - xb_[i];
                const double dn = norm2(dd);
                if (!(dn > 1e-3 * rhoend)) return;
                if (dedup) for (const Cand& c : cs) if (dist(c.x, xn) < 0.1 * rho) return;
                cs.push_back(Cand{xn, 0, -1, radius, q_.eval(dd) - q_.c, dn});
            };
            // the trust-region points are chosen AFTER the geometry points below (which reserve their slots first)
            double rmin = delta;
            auto add_tr_points = [&](int slots) {
                if (short_step) { push_tr(d1, delta, false); return; }
                const bool boundary = dn1 >= 0.999 * delta;
                // a search over the radius in one round: along the trust-region path when the step ends on the
                // boundary (delta, 2 delta, delta/2, ...), along the step itself when the model's minimiser is interior
                const double mult[6] = {1.0, 2.0, 0.5, 4.0, 0.25, 8.0};
                int added = 0;
                for (int t = 0; t < 6 && added < slots; ++t) {
                    const double r = mult[t] * (boundary ? delta : dn1);
                    if (r < 0.5 * rho) continue;
                    if (!boundary && mult[t] > 2.0) continue;
                    const size_t before = cs.size();
                    if (t == 0) push_tr(d1, r, true);
                    else if (boundary) push_tr(trust_step(q_, r, a, b, nullptr), r, true);
                    else { vec d(n); for (int i = 0; i < n; ++i) d[i] = mult[t] * d1[i]; push_tr(d, r, true); }
                    if (cs.size() > before) { rmin = std::min(rmin, r); ++added; }
                }
            };
            // replacements for the badly placed interpolation points, each chosen against the set as it will be.  After
            // a short step rho is about to shrink: place them for the next rho.
            int ngeom = 0;
            if (!(short_step && last_level)) {
                const double rho_g = short_step ? rho_next : rho;
                const double delta_g = short_step ? std::max(0.5 * rho, rho_next) : delta;
                const double far2 = std::max(4.0 * delta_g * delta_g, 100.0 * rho_g * rho_g);
                const std::vector<vec> Ysave = Y_;
                std::vector<char> tried(npt_, 0);
                const int reserve = short_step ? 1 : (room >= 6 ? 3 : room >= 3 ? 2 : 1);
                while ((int)cs.size() < room - reserve) {
                    int knew = -1; double dmax = far2;
                    for (int k = 0; k < npt_; ++k) {
                        if (k == kopt_ || tried[k]) continue;
                        const double sdist = dist(Y_[k], xb_); if (sdist * sdist > dmax) { dmax = sdist * sdist; knew = k; }
                    }
                    if (knew < 0) break;
                    tried[knew] = 1;
                    const double dk = std::sqrt(dmax);
                    const double adelt = std::max(std::min(0.1 * dk, delta_g), rho_g);
                    vec xn;
                    if (!geometry_point(knew, adelt, &xn)) continue;
                    bool dup = false;
                    for (const Cand& c : cs) if (dist(c.x, xn) < 0.1 * rho_g) dup = true;
                    for (int k = 0; k < npt_; ++k) if (k != knew && dist(Y_[k], xn) < 0.1 * rho_g) dup = true;
                    if (dup) continue;
                    Y_[knew] = xn;
                    if (!build_W()) { Y_[knew] = Ysave[knew]; build_W(); continue; }
                    cs.push_back(Cand{xn, 1, knew, adelt, 0.0, dist(xn, xb_)});
                    ++ngeom;
                }
                Y_ = Ysave; build_W();
            }
            add_tr_points(std::min(room - (int)cs.size(), short_step ? 1 : (room >= 6 ? 3 : room >= 3 ? 2 : 1)));
            // slots still free: a stencil around the point the model steps to -- the values the NEXT model wants if
            // the step is taken, and a direct check of it (the stencil brackets the minimiser) if it is not
            if ((int)cs.size() < room && !cs.empty() && cs.back().kind == 0 && !(short_step && last_level)) {
                const double rho_g = short_step ? rho_next : rho;
                const double hst = short_step ? rho_next : std::max(rho, 0.5 * dn1);
                vec xc(n); for (int i = 0; i < n; ++i) xc[i] = std::min(std::max(xb_[i] + d1[i], lo_[i]), up_[i]);
                auto push_st = [&](const vec& y0) {
                    vec y = y0; clampv(y);
                    for (const Cand& c : cs) if (dist(c.x, y) < 0.25 * hst) return;
                    for (int k = 0; k < npt_; ++k) if (dist(Y_[k], y) < std::max(0.1 * rho_g, 0.25 * hst)) return;
                    cs.push_back(Cand{y, 2, -1, hst, 0.0, dist(y, xb_)});
                };
                for (int sgn = 1; sgn >= -1 && (int)cs.size() < room; sgn -= 2)
                    for (int i = 0; i < n && (int)cs.size() < room; ++i) { vec y = xc; y[i] += sgn * hst; push_st(y); }
                const double hd = hst / std::sqrt(2.0);
                for (int si = 1; si >= -1 && (int)cs.size() < room; si -= 2)
                    for (int sj = 1; sj >= -1 && (int)cs.size() < room; sj -= 2)
                        for (int i = 0; i < n && (int)cs.size() < room; ++i)
                            for (int j = i + 1; j < n && (int)cs.size() < room; ++j) { vec y = xc; y[i] += si * hd; y[j] += sj * hd; push_st(y); }
            }
            bool improved_tr = false;
            if (!cs.empty()) {
                std::vector<vec> X; for (const Cand& c : cs) X.push_back(c.x);
                vec F;
                if (!eval_batch(X, &F)) return rc_;
                ++rounds;
                const double fopt0 = F_[kopt_];
                for (size_t i = 0; i < cs.size(); ++i)
                    if (cs[i].kind == 1) { replace(cs[i].knew, cs[i].x, F[i]); if (rc_) return rc_; }
                // trust-region points: worst first, so that the best one enters last and becomes the base point
                std::vector<int> tr;
                for (size_t i = 0; i < cs.size(); ++i) if (cs[i].kind != 1) tr.push_back((int)i);
                std::sort(tr.begin(), tr.end(), [&](int p, int q2) { return F[p] > F[q2] || (F[p] == F[q2] && p > q2); });
                int ibest = -1;
                for (int i : tr) {
                    if (cs[i].kind == 0 && (ibest < 0 || F[i] <= F[ibest])) ibest = i;
                    move_base_to_opt();
                    int knear = -1;
                    for (int k = 0; k < npt_; ++k) if (dist(Y_[k], cs[i].x) < 0.1 * rho) { knear = k; break; }
                    if (knear >= 0) {           // (nearly) on top of a point of the set: takes its place if it is better
                        if (F[i] < F_[knear]) {
                            const vec oldx = Y_[knear]; const double oldf = F_[knear];
                            Y_[knear] = cs[i].x; F_[knear] = F[i];
                            if (!refit(true)) { Y_[knear] = oldx; F_[knear] = oldf; build_W(); }
                            kopt_ = (int)(std::min_element(F_.begin(), F_.end()) - F_.begin());
                        }
                        continue;
                    }
                    const int knew = pick_replace(cs[i].x, F[i] < F_[kopt_], std::max(delta, rho));
                    if (knew >= 0) { replace(knew, cs[i].x, F[i]); if (rc_) return rc_; }
                }
                if (ibest >= 0 && !short_step) {
                    const Cand& cb = cs[ibest];
                    if (F[ibest] < fopt0) {
                        improved_tr = true;
                        const double ratio = cb.vquad < 0.0 ? (F[ibest] - fopt0) / cb.vquad : -1.0;
                        const double h = 0.5 * cb.radius;
                        if (ratio <= 0.1) delta = std::min(h, cb.dnorm);
                        else if (ratio <= 0.7) delta = std::max(h, cb.dnorm);
                        else delta = std::max(h, 2.0 * cb.dnorm);
                    } else delta = 0.5 * rmin;
                    if (delta <= 1.5 * rho) delta = rho;
                }
                if (F_[kopt_] < fopt0) improved_tr = true;      // a stencil or geometry point may be the one that improved
                if (o_.iprint > 1) fprintf(stderr, "bobyqa_batch:   %zu points (%d geometry) -> f=%.12g delta=%.3g\n", cs.size(), ngeom, F_[kopt_], delta);
            }
            bool reduce = false;
            if (short_step) reduce = true;
            else if (cs.empty()) { if (delta > rho) delta = rho; else reduce = true; }
            else if (!improved_tr && ngeom == 0 && rmin <= rho) reduce = true;
            if (reduce) {
                if (last_level) break;
                delta = std::max(0.5 * rho, rho_next);
                rho = rho_next;
                if (o_.iprint > 0) fprintf(stderr, "bobyqa_batch: rho -> %.3g  nf=%d rounds=%d f=%.12g\n", rho, nf_, rounds, F_[kopt_]);
            }
        }
        kopt_ = (int)(std::min_element(F_.begin(), F_.end()) - F_.begin());
        res->x = Y_[kopt_]; res->fval = F_[kopt_]; res->nfev = nf_; res->status = status; res->rounds = rounds;
        return MCML_OK;
    }

private:
    objective_fn none_;
    const objective_fn& f_;
    const batch_objective_fn* fb_ = nullptr;
    int n_, npt_ = 0, nf_ = 0, rc_ = 0, kopt_ = 0;
    vec lo_, up_;
    BobyqaOpts o_;
    vec x0_, xb_;
    std::vector<vec> Y_;
    vec F_;
    Quad q_;
    std::vector<double> Winv_; double sc_ = 1.0;   // inverse KKT matrix in coordinates (x - xb)/sc
    double fin_lo_ = HUGE_VAL, fin_hi_ = -HUGE_VAL; // range of the finite objective values seen (eval_batch)

    double eval(const vec& x)
    {
        double v = 0;
        int rc;
        if (fb_) {                      // batch mode: a batch of one (the rare rescue path)
            std::vector<vec> X(1, x); vec F;
            rc = (*fb_)(X, &F);
            if (!rc) v = F.empty() ? HUGE_VAL : F[0];
        } else rc = f_(x, &v);
        ++nf_;
        if (rc) { rc_ = rc; return 0; }
        if (v != v) v = HUGE_VAL;       // NaN objective: treat as +inf
        return v;
    }

    // one exchange: every point of X evaluated (by whoever owns it), all values returned
    bool eval_batch(const std::vector<vec>& X, vec* F)
    {
        F->assign(X.size(), 0.0);
        if (X.empty()) return true;
        int rc = (*fb_)(X, F);
        nf_ += (int)X.size();
        if (rc) { rc_ = rc; return false; }
        if (F->size() != X.size()) { rc_ = MCML_EINVAL; set_error("bobyqa_batch: objective returned %zu values for %zu points", F->size(), X.size()); return false; }
        // A point with no value (NaN, +inf: e.g. a covariance matrix that is not positive definite there) must stay
        // "worse than anything seen" without poisoning the interpolation models with infinities: it gets the largest
        // finite value seen so far plus ten times the spread of the finite values
        for (double v : *F) if (std::isfinite(v)) { fin_lo_ = std::min(fin_lo_, v); fin_hi_ = std::max(fin_hi_, v); }
        for (double& v : *F)
            if (!std::isfinite(v)) {
                v = (fin_hi_ >= fin_lo_) ? fin_hi_ + 10.0 * std::max(1.0, fin_hi_ - fin_lo_) : 1e30;
            }
        return true;
    }

    void move_base_to_opt()
    {
        kopt_ = (int)(std::min_element(F_.begin(), F_.end()) - F_.begin());
        vec d(n_); bool any = false;
        for (int i = 0; i < n_; ++i) { d[i] = Y_[kopt_][i] - xb_[i]; if (d[i] != 0.0) any = true; }
        if (any) { q_.shift(d); xb_ = Y_[kopt_]; build_W(); }
    }

    // KKT matrix of the minimum-Frobenius-norm interpolation problem, inverted
    bool build_W()
    {
        const int n = n_, m = npt_, N = m + n + 1;
        sc_ = 0;
        for (int k = 0; k < m; ++k) { double s = 0; for (int i = 0; i < n; ++i) { double t = Y_[k][i] - xb_[i]; s += t * t; } sc_ = std::max(sc_, std::sqrt(s)); }
        if (!(sc_ > 0)) return false;
        std::vector<double> S((size_t)n * m);
        for (int k = 0; k < m; ++k) for (int i = 0; i < n; ++i) S[i + (size_t)k * n] = (Y_[k][i] - xb_[i]) / sc_;
        std::vector<double> W((size_t)N * N, 0.0);
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) {
                double t = 0; for (int i = 0; i < n; ++i) t += S[i + (size_t)a * n] * S[i + (size_t)b * n];
                W[a + (size_t)b * N] = 0.5 * t * t;
            }
        for (int a = 0; a < m; ++a) {
            W[a + (size_t)m * N] = 1.0; W[m + (size_t)a * N] = 1.0;
            for (int i = 0; i < n; ++i) { W[a + (size_t)(m + 1 + i) * N] = S[i + (size_t)a * n]; W[(m + 1 + i) + (size_t)a * N] = S[i + (size_t)a * n]; }
        }
        if (!invert(W, N)) return false;
        Winv_.swap(W);
        return true;
    }

    // add to q_ the minimum-norm quadratic that interpolates the residuals f_k - q(y_k)
    bool refit(bool rebuild)
    {
        const int n = n_, m = npt_, N = m + n + 1;
        if (rebuild && !build_W()) return false;
        vec r(m);
        for (int k = 0; k < m; ++k) { vec d(n); for (int i = 0; i < n; ++i) d[i] = Y_[k][i] - xb_[i]; r[k] = F_[k] - q_.eval(d); }
        vec sol(N, 0.0);
        for (int a = 0; a < N; ++a) { double s = 0; for (int k = 0; k < m; ++k) s += Winv_[a + (size_t)k * N] * r[k]; sol[a] = s; }
        q_.c += sol[m];
        for (int i = 0; i < n; ++i) q_.g[i] += sol[m + 1 + i] / sc_;
        for (int k = 0; k < m; ++k) {
            const double lam = sol[k] / (sc_ * sc_ * sc_ * sc_);
            if (lam == 0.0) continue;
            for (int j = 0; j < n; ++j) { double sj = Y_[k][j] - xb_[j]; if (sj == 0.0) continue;
                for (int i = 0; i < n; ++i) q_.H[i + (size_t)j * n] += lam * (Y_[k][i] - xb_[i]) * sj; }
        }
        return true;
    }

    // Lagrange polynomial t as a quadratic about xb
    Quad lagrange(int t) const
    {
        const int n = n_, m = npt_, N = m + n + 1;
        Quad l; l.init(n);
        l.c = Winv_[m + (size_t)t * N];
        for (int i = 0; i < n; ++i) l.g[i] = Winv_[(m + 1 + i) + (size_t)t * N] / sc_;
        for (int k = 0; k < m; ++k) {
            const double lam = Winv_[k + (size_t)t * N] / (sc_ * sc_ * sc_ * sc_);
            if (lam == 0.0) continue;
            for (int j = 0; j < n; ++j) { double sj = Y_[k][j] - xb_[j]; if (sj == 0.0) continue;
                for (int i = 0; i < n; ++i) l.H[i + (size_t)j * n] += lam * (Y_[k][i] - xb_[i]) * sj; }
        }
        return l;
    }

    // which point leaves when xnew enters: largest |l_t(xnew)| weighted by distance
    int pick_replace(const vec& xnew, bool improved, double delta)
    {
        const int n = n_;
        vec d(n); for (int i = 0; i < n; ++i) d[i] = xnew[i] - xb_[i];
        int best = -1; double bw = 0;
        const double delsq = delta * delta;
        for (int t = 0; t < npt_; ++t) {
            if (t == kopt_ && !improved) continue;
            Quad l = lagrange(t);
            const double lv = l.eval(d);
            double dist = 0;
            const vec& ref = improved ? xnew : Y_[kopt_];
            for (int i = 0; i < n; ++i) { double s = Y_[t][i] - ref[i]; dist += s * s; }
            double w = std::max(1.0, (dist / delsq) * (dist / delsq));
            double sc = w * lv * lv;
            if (sc > bw) { bw = sc; best = t; }
        }
        if (best < 0 && improved) best = kopt_;
        return best;
    }

    void replace(int knew, const vec& xnew, double fnew)
    {
        vec oldx = Y_[knew]; double oldf = F_[knew];
        Y_[knew] = xnew; F_[knew] = fnew;
        if (!refit(true)) {            // degenerate geometry: keep the old point unless the new one is better
            if (fnew < oldf) { reinit_around_best(); }
            else { Y_[knew] = oldx; F_[knew] = oldf; build_W(); }
        }
        kopt_ = (int)(std::min_element(F_.begin(), F_.end()) - F_.begin());
    }

    // RESCUE's job: rebuild a well-poised set around the best point
    void reinit_around_best()
    {
        kopt_ = (int)(std::min_element(F_.begin(), F_.end()) - F_.begin());
        vec x = Y_[kopt_]; double fx = F_[kopt_];
        double rad = 0;
        for (int k = 0; k < npt_; ++k) { double s = 0; for (int i = 0; i < n_; ++i) { double t = Y_[k][i] - x[i]; s += t * t; } rad = std::max(rad, std::sqrt(s)); }
        if (!(rad > 0)) rad = 1e-3;
        rad = std::min(rad, 1.0);
        Y_[0] = x; F_[0] = fx;
        for (int k = 1; k < npt_; ++k) {
            int i = (k - 1) % n_; bool second = (k - 1) >= n_;
            vec y = x;
            double step = second ? -rad : rad;
            if (y[i] + step > up_[i]) step = -std::fabs(step) * (second ? 2 : 1);
            if (y[i] + step < lo_[i]) step = std::fabs(step) * (second ? 2 : 1);
            y[i] = std::min(std::max(y[i] + step, lo_[i]), up_[i]);
            Y_[k] = y; F_[k] = eval(y);
            if (rc_) return;
        }
        kopt_ = 0; xb_ = x; q_.init(n_);
        refit(true);
    }

    // ALTMOV's job: a point within adelt of xopt (and the bounds) that makes |l_knew| large
    bool geometry_point(int knew, double adelt, vec* out)
    {
        const int n = n_;
        Quad l = lagrange(knew);
        vec a(n), b(n);
        for (int i = 0; i < n; ++i) { a[i] = lo_[i] - xb_[i]; b[i] = up_[i] - xb_[i]; }
        Quad lneg = l; lneg.c = -l.c; for (auto& v : lneg.g) v = -v; for (auto& v : lneg.H) v = -v;
        vec d1 = trust_step(l, adelt, a, b, nullptr), d2 = trust_step(lneg, adelt, a, b, nullptr);
        double v1 = std::fabs(l.eval(d1)), v2 = std::fabs(l.eval(d2));
        vec best = v1 >= v2 ? d1 : d2; double bv = std::max(v1, v2);
        // also the straight line towards (and away from) the outgoing point, as ALTMOV tries
        vec dir(n); double dn = 0; for (int i = 0; i < n; ++i) { dir[i] = Y_[knew][i] - xb_[i]; dn += dir[i] * dir[i]; }
        dn = std::sqrt(dn);
        if (dn > 0) for (int sgn = -1; sgn <= 1; sgn += 2) {
            vec dd(n); for (int i = 0; i < n; ++i) dd[i] = std::min(std::max(sgn * adelt * dir[i] / dn, a[i]), b[i]);
            double v = std::fabs(l.eval(dd)); if (v > bv) { bv = v; best = dd; }
        }
        if (!(bv > 1e-12) || !(norm2(best) > 0)) return false;
        out->resize(n);
        for (int i = 0; i < n; ++i) (*out)[i] = std::min(std::max(xb_[i] + best[i], lo_[i]), up_[i]);
        return true;
    }
};

}  // namespace

int bobyqa(const objective_fn& f, const std::vector<double>& x0, const std::vector<double>& lower,
           const std::vector<double>& upper, const BobyqaOpts& opts, BobyqaResult* res)
{
    MCML_REQUIRE(res && !x0.empty() && lower.size() == x0.size() && upper.size() == x0.size(), "bobyqa: bad arguments");
    if (x0.size() == 1) {
        // BOBYQA needs n >= 2 (npt in [n+2, (n+1)(n+2)/2] is empty for n = 1): embed in two dimensions
        objective_fn f2 = [&](const std::vector<double>& x, double* v) { std::vector<double> x1(1, x[0]); int rc = f(x1, v); if (!rc) *v += x[1] * x[1]; return rc; };
        std::vector<double> x2{x0[0], 0.0}, l2{lower[0], -1.0}, u2{upper[0], 1.0};
        BobyqaResult r2;
        BobyqaOpts o2 = opts;
        if (!(o2.rhobeg > 0)) { double a = std::fabs(x0[0]); o2.rhobeg = std::min(0.95, 0.2 * a); if (!(o2.rhobeg > 0)) o2.rhobeg = 0.1; }
        Bobyqa b(f2, x2, l2, u2, o2);
        int rc = b.run(&r2);
        if (rc) return rc;
        res->x.assign(1, r2.x[0]); res->nfev = r2.nfev; res->status = r2.status;
        return f(res->x, &res->fval);
    }
    BobyqaOpts o = opts;
    if (const char* t = getenv("GLMMR_MCML_BOBYQA_TRACE")) o.iprint = atoi(t);
    Bobyqa b(f, x0, lower, upper, o);
    return b.run(res);
}

int bobyqa_batch(const batch_objective_fn& f, const std::vector<double>& x0, const std::vector<double>& lower,
                 const std::vector<double>& upper, const BobyqaOpts& opts, int width, BobyqaResult* res)
{
    MCML_REQUIRE(res && !x0.empty() && lower.size() == x0.size() && upper.size() == x0.size() && width >= 1, "bobyqa_batch: bad arguments");
    BobyqaOpts o = opts;
    if (const char* t = getenv("GLMMR_MCML_BOBYQA_TRACE")) o.iprint = atoi(t);
    if (x0.size() == 1) {
        // as bobyqa(): one parameter is embedded in two dimensions
        batch_objective_fn f2 = [&](const std::vector<std::vector<double>>& X, std::vector<double>* F) {
            std::vector<std::vector<double>> X1; for (const auto& x : X) X1.push_back(std::vector<double>(1, x[0]));
            int rc = f(X1, F);
            if (!rc && F->size() == X.size()) for (size_t i = 0; i < X.size(); ++i) (*F)[i] += X[i][1] * X[i][1];
            return rc; };
        std::vector<double> x2{x0[0], 0.0}, l2{lower[0], -1.0}, u2{upper[0], 1.0};
        if (!(o.rhobeg > 0)) { double a = std::fabs(x0[0]); o.rhobeg = std::min(0.95, 0.2 * a); if (!(o.rhobeg > 0)) o.rhobeg = 0.1; }
        BobyqaResult r2;
        Bobyqa b(f2, x2, l2, u2, o);
        int rc = width == 1 ? b.run(&r2) : b.run_batch(width, &r2);
        if (width == 1) r2.rounds = r2.nfev;
        if (rc) return rc;
        res->x.assign(1, r2.x[0]); res->nfev = r2.nfev; res->status = r2.status; res->rounds = r2.rounds;
        res->fval = r2.fval - r2.x[1] * r2.x[1];
        return MCML_OK;
    }
    Bobyqa b(f, x0, lower, upper, o);
    if (width == 1) {       // nothing to run side by side: the sequential schedule, one point per exchange
        int rc = b.run(res);
        res->rounds = res->nfev;
        return rc;
    }
    return b.run_batch(width, res);
}

// ---------------------------------------------------------------- finite differences
// R's optim.c fmingr (numerical branch) as rminqa's Functor::Gradient restates it
int fd_gradient(const objective_fn& f, const std::vector<double>& p, const std::vector<double>& ndeps,
                bool usebounds, const std::vector<double>& lower, const std::vector<double>& upper,
                std::vector<double>* df)
{
    const int n = (int)p.size();
    df->assign(n, 0.0);
    std::vector<double> x = p;
    for (int i = 0; i < n; ++i) {
        double val1, val2;
        if (!usebounds) {
            const double eps = ndeps[i];
            x[i] = p[i] + eps; MCML_TRY(f(x, &val1));
            x[i] = p[i] - eps; MCML_TRY(f(x, &val2));
            (*df)[i] = (val1 - val2) / (2 * eps);
        } else {
            double epsused = ndeps[i], eps = ndeps[i];
            double tmp = p[i] + eps;
            if (tmp > upper[i]) { tmp = upper[i]; epsused = tmp - p[i]; }
            x[i] = tmp; MCML_TRY(f(x, &val1));
            tmp = p[i] - eps;
            if (tmp < lower[i]) { tmp = lower[i]; eps = p[i] - tmp; }
            x[i] = tmp; MCML_TRY(f(x, &val2));
            (*df)[i] = (val1 - val2) / (epsused + eps);
        }
        x[i] = p[i];
    }
    return MCML_OK;
}

// R's optimhess
int fd_hessian(const objective_fn& f, const std::vector<double>& p, const std::vector<double>& ndeps,
               bool usebounds, const std::vector<double>& lower, const std::vector<double>& upper,
               std::vector<double>* H)
{
    const int n = (int)p.size();
    H->assign((size_t)n * n, 0.0);
    std::vector<double> dpar = p, df1, df2;
    for (int i = 0; i < n; ++i) {
        const double eps = ndeps[i];
        dpar[i] = dpar[i] + eps;
        MCML_TRY(fd_gradient(f, dpar, ndeps, usebounds, lower, upper, &df1));
        dpar[i] = dpar[i] - 2 * eps;
        MCML_TRY(fd_gradient(f, dpar, ndeps, usebounds, lower, upper, &df2));
        for (int j = 0; j < n; ++j) (*H)[i + (size_t)j * n] = (df1[j] - df2[j]) / (2 * eps);
        dpar[i] = dpar[i] + eps;
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) {
            double t = 0.5 * ((*H)[i + (size_t)j * n] + (*H)[j + (size_t)i * n]);
            (*H)[i + (size_t)j * n] = t; (*H)[j + (size_t)i * n] = t;
        }
    return MCML_OK;
}

}  // namespace mcml
