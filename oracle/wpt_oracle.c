/* ORACLE (test infrastructure only) -- plain C restatement of the wavelet-packet
 * front end.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product never does.
 *
 * Restates reference src/audiofakedetect/wavelet_math.py:167-220 (packet tree,
 * frequency-order gather, log-power, sign channel) and the ptwt analysis step it calls
 * at wavelet_math.py:182,192 (third party, unpinned; published algorithm: reflect pad
 * (L-2, L-2 + (n odd)), stride-2 correlation with the flipped taps).
 * Parity pin: see oracle/wpt_oracle.py header ("parity unpinned at the ptwt boundary").
 *
 * Written independently of the numpy version: explicit reflect indexing instead of a
 * padded copy, depth-first recursion instead of level-by-level dictionaries.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static long reflect_index(long j, long n) {
    /* whole-sample symmetric extension without edge repeat: x[-1]=x[1], x[n]=x[n-2] */
    if (j < 0) j = -j;
    if (j >= n) j = 2 * (n - 1) - j;
    return j;
}

static long child_len(long n, long L) { return (n + L - 2 + (n & 1)) / 2; }

static void analysis(const double *x, long n, const double *lo, const double *hi, long L,
                     double *ca, double *cd) {
    long nout = child_len(n, L);
    for (long i = 0; i < nout; ++i) {
        double sa = 0.0, sd = 0.0;
        for (long m = 0; m < L; ++m) {
            double v = x[reflect_index(2 * i + 1 - m, n)];
            sa += lo[m] * v;
            sd += hi[m] * v;
        }
        ca[i] = sa;
        cd[i] = sd;
    }
}

/* depth-first: node at `depth` with gray-ordered frequency index `f` */
static void recurse(const double *x, long n, const double *lo, const double *hi, long L,
                    int depth, int level, long f, double *out, long T) {
    if (depth == level) {
        memcpy(out + f * T, x, (size_t)n * sizeof(double));
        return;
    }
    long nout = child_len(n, L);
    double *ca = (double *)malloc((size_t)nout * sizeof(double));
    double *cd = (double *)malloc((size_t)nout * sizeof(double));
    analysis(x, n, lo, hi, L, ca, cd);
    /* frequency order: children of an even-index node are (a,d), of an odd one (d,a) */
    long fa = 2 * f + (f & 1);
    long fd = 2 * f + 1 - (f & 1);
    recurse(ca, nout, lo, hi, L, depth + 1, level, fa, out, T);
    recurse(cd, nout, lo, hi, L, depth + 1, level, fd, out, T);
    free(ca);
    free(cd);
}

/* x [B][N] double -> out [B][P][T] double (P = 2^level, frequency order). returns T. */
long wpt_oracle_nodes(const double *x, long B, long N, const double *dec_lo, long L,
                      int level, double *out) {
    double *hi = (double *)malloc((size_t)L * sizeof(double));
    for (long k = 0; k < L; ++k) hi[k] = ((k + 1) % 2 ? -1.0 : 1.0) * dec_lo[L - 1 - k];
    long T = N;
    for (int l = 0; l < level; ++l) T = child_len(T, L);
    long P = 1L << level;
    if (out) {
        for (long b = 0; b < B; ++b)
            recurse(x + b * N, N, dec_lo, hi, L, 0, level, 0, out + b * P * T, T);
    }
    free(hi);
    return T;
}

/* features: out [B][C][P][T] (logical layout), C = 1 or 2 */
long wpt_oracle_features(const double *x, long B, long N, const double *dec_lo, long L,
                         int level, int log_scale, int loss_less, double power,
                         double *out) {
    long T = wpt_oracle_nodes(x, B, N, dec_lo, L, level, NULL);
    long P = 1L << level;
    double *nodes = (double *)malloc((size_t)(B * P * T) * sizeof(double));
    wpt_oracle_nodes(x, B, N, dec_lo, L, level, nodes);
    int C = (log_scale && loss_less) ? 2 : 1;
    for (long b = 0; b < B; ++b)
        for (long i = 0; i < P * T; ++i) {
            double v = nodes[b * P * T + i];
            if (log_scale) {
                out[(b * C) * P * T + i] = log(pow(fabs(v), power) + 1e-12);
                if (loss_less) out[(b * C + 1) * P * T + i] = (v < 0) ? -1.0 : 1.0;
            } else {
                out[b * P * T + i] = v;
            }
        }
    free(nodes);
    return T;
}
