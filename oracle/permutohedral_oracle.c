/*
 * Plain-C restatement of the permutohedral-lattice Gaussian filter used by probreg's FilterReg
 * (TEST INFRASTRUCTURE - the parity oracle for the HIP lattice kernels).
 *
 * Restates, in scalar float32 arithmetic, what the reference executes on x86-64:
 *   /root/reference/third_party/permutohedral/permutohedral.cpp
 *     :140-325  Permutohedral::init, SSE build  (elevate :197-204, round-half-even :207-218,
 *               rank :221-232, wrap :235-241, barycentric :244-262, keys + hash :264-274,
 *               blur neighbours :300-324)
 *     :482-533  seqCompute  (used for <= 2 value channels: blur evaluates 0.5*(n1+n2) in double)
 *     :535-596  sseCompute  (used for  > 2 value channels: everything in float)
 *     :603-616  compute(): `start` is dropped, `alpha = 1/(1+2^-d)`.
 * Vertex ids are handed out in first-touch order over (point, remainder) exactly like the reference's
 * HashTable::find(create=true) (:95-125), so offsets are comparable one to one with oracle/_ref.
 * Checked against the vendored reference itself (oracle/_ref/libpermuto_ref.so) in
 * tests/test_oracle_filterreg.py.  Parity status: PINNED.
 *
 * Build: gcc -O2 -shared -fPIC permutohedral_oracle.c -o libpermuto_oracle.so -lm
 * (no -ffast-math, no FMA contraction: -O2 on x86-64-v1 keeps float ops as written)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXD 64 /* feature lattices: FPFH is d = 33 (probreg/features.py:28-51) */

typedef struct {
    int n, d, m, with_blur;
    int* offset;       /* [n][d+1] */
    float* bary;       /* [n][d+1] */
    short* keys;       /* [m][d] */
    int* nb1;          /* [d+1][m] */
    int* nb2;
    /* hash */
    int64_t cap;
    int* table;
} lattice_t;

static uint64_t hash_key(const short* k, int d) {
    uint64_t r = 0;
    for (int i = 0; i < d; ++i) {
        r += (uint64_t)(int64_t)k[i];
        r *= 1664525u;
    }
    return r;
}

static int table_find(lattice_t* L, const short* k, int create) {
    uint64_t h = hash_key(k, L->d) % (uint64_t)L->cap;
    for (;;) {
        int e = L->table[h];
        if (e < 0) {
            if (!create) return -1;
            memcpy(L->keys + (size_t)L->m * L->d, k, sizeof(short) * L->d);
            L->table[h] = L->m;
            return L->m++;
        }
        if (memcmp(L->keys + (size_t)e * L->d, k, sizeof(short) * L->d) == 0) return e;
        if (++h == (uint64_t)L->cap) h = 0;
    }
}

void* permuto_oracle_create(const float* feat /* n x d row-major */, int n, int d, int with_blur) {
    if (d < 1 || d > MAXD) return 0;
    lattice_t* L = (lattice_t*)calloc(1, sizeof(lattice_t));
    L->n = n; L->d = d; L->with_blur = with_blur;
    const int d1 = d + 1;
    L->offset = (int*)malloc(sizeof(int) * (size_t)n * d1);
    L->bary = (float*)malloc(sizeof(float) * (size_t)n * d1);
    L->keys = (short*)malloc(sizeof(short) * ((size_t)n * d1 + 1) * d);
    L->cap = 4 * (int64_t)n * d1 + 16;
    L->table = (int*)malloc(sizeof(int) * (size_t)L->cap);
    for (int64_t i = 0; i < L->cap; ++i) L->table[i] = -1;

    short canonical[(MAXD + 1) * (MAXD + 1)];
    for (int i = 0; i <= d; ++i) {
        for (int j = 0; j <= d - i; ++j) canonical[i * d1 + j] = (short)i;
        for (int j = d - i + 1; j <= d; ++j) canonical[i * d1 + j] = (short)(i - d1);
    }
    /* :180-183  float inv_std_dev; scale_factor = float(1.0/sqrt((i+2)(i+1)) * inv_std_dev) */
    const float inv_std_dev = with_blur ? (float)(sqrt(2.0 / 3.0) * d1) : (float)(sqrt(1.0 / 6.0) * d1);
    float scale[MAXD];
    for (int i = 0; i < d; ++i) scale[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * (double)inv_std_dev);
    const float invd1 = 1.0f / (float)d1, fd1 = (float)d1;

    for (int k = 0; k < n; ++k) {
        const float* f = feat + (size_t)k * d;
        float elevated[MAXD + 1], rem0[MAXD + 1], rank[MAXD + 1], bar[MAXD + 2];
        volatile float sm = 0.0f; /* volatile: keep every float op individually rounded */
        for (int j = d; j > 0; --j) {
            const float cf = f[j - 1] * scale[j - 1];
            const float jc = (float)j * cf;
            elevated[j] = sm - jc;
            sm = sm + cf;
        }
        elevated[0] = sm;
        float sum = 0.0f;
        for (int i = 0; i <= d; ++i) {
            float v = invd1 * elevated[i];
            v = nearbyintf(v); /* round half to even, like _mm_cvtps_epi32 / _MM_FROUND_TO_NEAREST_INT */
            rem0[i] = v * fd1;
            sum += v;
        }
        for (int i = 0; i <= d; ++i) rank[i] = 0.0f;
        for (int i = 0; i < d; ++i) {
            const float di = elevated[i] - rem0[i];
            for (int j = i + 1; j <= d; ++j) {
                const float dj = elevated[j] - rem0[j];
                if (di < dj) rank[i] += 1.0f; else rank[j] += 1.0f;
            }
        }
        for (int i = 0; i <= d; ++i) {
            rank[i] += sum;
            if (rank[i] < 0.0f) { rank[i] += fd1; rem0[i] += fd1; }
            else if (rank[i] >= fd1) { rank[i] -= fd1; rem0[i] -= fd1; }
        }
        for (int i = 0; i <= d + 1; ++i) bar[i] = 0.0f;
        for (int i = 0; i <= d; ++i) {
            const float v = (elevated[i] - rem0[i]) * invd1;
            const int p = d - (int)rank[i];
            bar[p] += v;
            bar[p + 1] -= v;
        }
        bar[0] += 1.0f + bar[d + 1];
        for (int r = 0; r <= d; ++r) {
            short key[MAXD];
            for (int i = 0; i < d; ++i) key[i] = (short)(rem0[i] + (float)canonical[r * d1 + (int)rank[i]]);
            L->offset[(size_t)k * d1 + r] = table_find(L, key, 1);
            L->bary[(size_t)k * d1 + r] = bar[r];
        }
    }
    if (with_blur) {
        L->nb1 = (int*)malloc(sizeof(int) * (size_t)d1 * L->m);
        L->nb2 = (int*)malloc(sizeof(int) * (size_t)d1 * L->m);
        const int mm = L->m;
        for (int j = 0; j <= d; ++j)
            for (int i = 0; i < mm; ++i) {
                const short* key = L->keys + (size_t)i * d;
                short n1[MAXD + 1], n2[MAXD + 1];
                for (int k = 0; k < d; ++k) { n1[k] = (short)(key[k] - 1); n2[k] = (short)(key[k] + 1); }
                if (j < d) { n1[j] = (short)(key[j] + d); n2[j] = (short)(key[j] - d); }
                L->nb1[(size_t)j * mm + i] = table_find(L, n1, 0);
                L->nb2[(size_t)j * mm + i] = table_find(L, n2, 0);
            }
    }
    return L;
}

int permuto_oracle_size(void* h) { return ((lattice_t*)h)->m; }

void permuto_oracle_get(void* h, int* offset, float* bary, short* keys) {
    lattice_t* L = (lattice_t*)h;
    const int d1 = L->d + 1;
    if (offset) memcpy(offset, L->offset, sizeof(int) * (size_t)L->n * d1);
    if (bary) memcpy(bary, L->bary, sizeof(float) * (size_t)L->n * d1);
    if (keys) memcpy(keys, L->keys, sizeof(short) * (size_t)L->m * L->d);
}

/* values: n x ch row-major -> out n x ch.  ch <= 2 follows seqCompute, ch > 2 follows sseCompute
 * (the SSE path pads each vertex to multiples of 4 floats; padding lanes never mix, so plain
 * per-channel float arithmetic is the same). */
void permuto_oracle_filter(void* h, const float* in, int ch, float* out) {
    lattice_t* L = (lattice_t*)h;
    const int d = L->d, d1 = d + 1, m = L->m, n = L->n;
    const int seq = ch <= 2;
    float* val = (float*)calloc((size_t)(m + 2) * ch, sizeof(float));
    float* nval = (float*)calloc((size_t)(m + 2) * ch, sizeof(float));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= d; ++j) {
            const int o = L->offset[(size_t)i * d1 + j] + 1;
            const float w = L->bary[(size_t)i * d1 + j];
            for (int k = 0; k < ch; ++k) {
                const float p = w * in[(size_t)i * ch + k];
                val[(size_t)o * ch + k] += p;
            }
        }
    if (L->with_blur) {
        for (int j = 0; j <= d; ++j) {
            for (int i = 0; i < m; ++i) {
                const float* ov = val + (size_t)(i + 1) * ch;
                float* nv = nval + (size_t)(i + 1) * ch;
                const float* a = val + (size_t)(L->nb1[(size_t)j * m + i] + 1) * ch;
                const float* b = val + (size_t)(L->nb2[(size_t)j * m + i] + 1) * ch;
                for (int k = 0; k < ch; ++k) {
                    if (seq) {
                        const float s = a[k] + b[k];               /* float add (:510) */
                        nv[k] = (float)((double)ov[k] + 0.5 * (double)s); /* 0.5 is a double literal */
                    } else {
                        const float s = a[k] + b[k];
                        const float hs = 0.5f * s;
                        nv[k] = ov[k] + hs;
                    }
                }
            }
            float* t = val; val = nval; nval = t;
        }
    }
    const float alpha = 1.0f / (1.0f + powf(2.0f, (float)-d));
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < ch; ++k) out[(size_t)i * ch + k] = 0.0f;
        for (int j = 0; j <= d; ++j) {
            const int o = L->offset[(size_t)i * d1 + j] + 1;
            const float w = L->bary[(size_t)i * d1 + j];
            for (int k = 0; k < ch; ++k) {
                if (seq) {
                    const float p = w * val[(size_t)o * ch + k];
                    const float q = p * alpha;
                    out[(size_t)i * ch + k] += q;
                } else {
                    const float wa = w * alpha;
                    const float p = wa * val[(size_t)o * ch + k];
                    out[(size_t)i * ch + k] += p;
                }
            }
        }
    }
    free(val);
    free(nval);
}

void permuto_oracle_destroy(void* h) {
    lattice_t* L = (lattice_t*)h;
    if (!L) return;
    free(L->offset); free(L->bary); free(L->keys); free(L->nb1); free(L->nb2); free(L->table);
    free(L);
}
