// std_stream.hpp -- the CIGAR standardisation of realign_read (reference src/bam.pyx:65-78, src/cig.pyx:102-192) as ONE
// streaming pass over the RUNS of an alignment, written once and compiled for the host (hostio.hpp / npore_api.cpp) and
// for the gfx950 kernel that standardises a batch on the device (one wavefront per read, kernels.hpp standardize_kernel).
//
// npore_amd/cig.py states the same five steps as five passes over run lists; here each step is a stage with O(1) state
// that hands every run it can no longer change to the next stage:
//   A  push_indels_left('D', reference)   a deletion run moves left through the match run right before it while the
//                                         reference it deletes stays the same (ref[x] == ref[x + k]); only that match
//                                         run -- the last of the stage's output -- can still change, and when it is used
//                                         up the deletions join the run before it: a window of TWO runs is held back
//   B  push_inss_thru_dels                every maximal block of I / D runs becomes 'I..I D..D': two counters
//   C  push_indels_left('I', read)        like A
//   D  push_inss_thru_dels                like B
//   E  'ID' -> 'M'                        one pair per block: a pending insertion run and one held run (the 'M' may merge
//                                         with the match runs either side)
// The stages see exactly the runs the passes see, in the same order, so the result is the same run for run
// (tests/test_host_logic.py: both forms against the per-op restatement of the reference on random alignments over
// low-complexity sequences, and on the golden reads).  No allocation, no recursion, no containers.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NPORE_STD_HD __host__ __device__ __forceinline__
#else
#define NPORE_STD_HD inline
#endif

namespace npore {

enum : uint32_t { SOP_M = 0, SOP_I = 1, SOP_D = 2, SOP_NONE = 3 };

// How far an indel run of length k at position p of the consumed sequence s_ slides left through the m matches in front
// of it: the number of t = 0, 1, ... < m with s_[p - t - 1] == s_[p - t - 1 + k] before the first that differs (positions
// at or beyond s_len do not match: the reference indexes unchecked there).  The host walks position by position; the
// device kernel (one wavefront per read) compares 64 positions per round trip (kernels.hpp WaveProbe).
struct ScalarProbe {
    template <class Int>
    NPORE_STD_HD Int operator()(const uint8_t *s_, Int p, Int k, Int m, Int s_len) const
    {
        Int s = 0;
        while (s < m && p - s - 1 + k < s_len && s_[p - s - 1] == s_[p - s - 1 + k]) s++;
        return s;
    }
};

// Sink: void operator()(uint32_t op, Int len) -- called once per run of the result, never twice in a row with the
// same op.  `ref` / `seq`: the bases the alignment pairs (any encoding: only compared for equality).  Int: the type of
// lengths and positions (the device kernel uses 32 bits -- a read has fewer than 2^31 ops -- to halve its registers).
template <class Sink, class Int = int64_t, class Probe = ScalarProbe>
struct StdStream {
    Sink &sink;
    Probe probe;
    const uint8_t *ref, *seq;
    Int ref_len, seq_len;

    NPORE_STD_HD StdStream(Sink &s, const uint8_t *r, Int rl, const uint8_t *q, Int ql)
        : sink(s), ref(r), seq(q), ref_len(rl), seq_len(ql) {}

    // ---- stage E: 'ID' -> 'M', then the sink (a held run so that equal neighbours merge)
    uint32_t e_op = SOP_NONE;
    Int e_len = 0;
    Int e_pend_i = 0;                                  // an insertion run waiting to see whether a deletion run follows
    NPORE_STD_HD void e_push(uint32_t op, Int len)
    {
        if (len <= 0) return;
        if (e_op == op) { e_len += len; return; }
        if (e_op != SOP_NONE) sink(e_op, e_len);
        e_op = op;
        e_len = len;
    }
    NPORE_STD_HD void e_feed(uint32_t op, Int len)
    {
        if (e_pend_i) {
            const Int a = e_pend_i;
            e_pend_i = 0;
            if (op == SOP_D) {                             // 'I..I D..D' -> 'I..(a-1) M D..(b-1)'
                e_push(SOP_I, a - 1);
                e_push(SOP_M, 1);
                e_push(SOP_D, len - 1);
                return;
            }
            e_push(SOP_I, a);
        }
        if (op == SOP_I) e_pend_i = len;
        else e_push(op, len);
    }
    NPORE_STD_HD void e_end()
    {
        if (e_pend_i) { e_push(SOP_I, e_pend_i); e_pend_i = 0; }
        if (e_op != SOP_NONE) sink(e_op, e_len);
        e_op = SOP_NONE;
    }

    // ---- stages B / D: insertions before deletions within a block
    struct Block {
        Int ni = 0, nd = 0;
    };
    Block blk_b, blk_d;
    template <int WHICH>
    NPORE_STD_HD void blk_out(uint32_t op, Int len)
    {
        if (len <= 0) return;
        if (WHICH == 0) c_feed(op, len);
        else e_feed(op, len);
    }
    template <int WHICH>
    NPORE_STD_HD void blk_feed(uint32_t op, Int len)
    {
        Block &k = WHICH == 0 ? blk_b : blk_d;
        if (op == SOP_M) {
            blk_out<WHICH>(SOP_I, k.ni);
            blk_out<WHICH>(SOP_D, k.nd);
            k.ni = k.nd = 0;
            blk_out<WHICH>(SOP_M, len);
        } else if (op == SOP_I) {
            k.ni += len;
        } else {
            k.nd += len;
        }
    }
    template <int WHICH>
    NPORE_STD_HD void blk_end()
    {
        Block &k = WHICH == 0 ? blk_b : blk_d;
        blk_out<WHICH>(SOP_I, k.ni);
        blk_out<WHICH>(SOP_D, k.nd);
        k.ni = k.nd = 0;
    }

    // ---- stages A / C: push the runs of one indel kind left
    struct Push {
        uint32_t op0 = SOP_NONE, op1 = SOP_NONE;          // the two runs held back (op1 the later one)
        Int len0 = 0, len1 = 0;
        Int p = 0;                                     // position in the consumed sequence of the next op
    };
    Push push_a, push_c;
    template <int WHICH>
    NPORE_STD_HD void push_out(uint32_t op, Int len)
    {
        if (WHICH == 0) blk_feed<0>(op, len);
        else blk_feed<1>(op, len);
    }
    template <int WHICH>
    NPORE_STD_HD void push_run(uint32_t op, Int len)  // append to the stage's output, merging with its last run
    {
        Push &w = WHICH == 0 ? push_a : push_c;
        if (len <= 0) return;
        if (w.op1 != SOP_NONE) {
            if (w.op1 == op) { w.len1 += len; return; }
            if (w.op0 != SOP_NONE) push_out<WHICH>(w.op0, w.len0);
            w.op0 = w.op1; w.len0 = w.len1;
            w.op1 = op; w.len1 = len;
        } else if (w.op0 != SOP_NONE) {
            if (w.op0 == op) { w.len0 += len; return; }
            w.op1 = op; w.len1 = len;
        } else {
            w.op0 = op; w.len0 = len;
        }
    }
    template <int WHICH>
    NPORE_STD_HD void push_feed(uint32_t op, Int len)
    {
        Push &w = WHICH == 0 ? push_a : push_c;
        const uint32_t push_op = WHICH == 0 ? SOP_D : SOP_I;
        const uint8_t *s_ = WHICH == 0 ? ref : seq;
        const Int s_len = WHICH == 0 ? ref_len : seq_len;
        if (op != push_op) {
            push_run<WHICH>(op, len);
            if (op == SOP_M) w.p += len;
            return;
        }
        // the last run of the output, if it is a match run: where it lies in the window
        const bool last1 = w.op1 != SOP_NONE;
        const uint32_t lop = last1 ? w.op1 : w.op0;
        const Int k = len, m = (lop == SOP_M) ? (last1 ? w.len1 : w.len0) : 0;
        const Int p = w.p;
        const Int s = m > 0 ? probe(s_, p, k, m, s_len) : 0;
        if (s) {
            if (last1) { w.len1 -= s; if (w.len1 == 0) w.op1 = SOP_NONE; }
            else { w.len0 -= s; if (w.len0 == 0) w.op0 = SOP_NONE; }
        }
        push_run<WHICH>(push_op, k);
        push_run<WHICH>(SOP_M, s);
        w.p += k;
    }
    template <int WHICH>
    NPORE_STD_HD void push_end()
    {
        Push &w = WHICH == 0 ? push_a : push_c;
        if (w.op0 != SOP_NONE) push_out<WHICH>(w.op0, w.len0);
        if (w.op1 != SOP_NONE) push_out<WHICH>(w.op1, w.len1);
        w.op0 = w.op1 = SOP_NONE;
    }
    NPORE_STD_HD void c_feed(uint32_t op, Int len) { push_feed<1>(op, len); }

    // ---- the source side: runs of the alignment in read order ('=', 'X', 'M' all SOP_M); equal neighbours are merged here
    uint32_t in_op = SOP_NONE;
    Int in_len = 0;
    NPORE_STD_HD void feed(uint32_t op, Int len)
    {
        if (len <= 0) return;
        if (op == in_op) { in_len += len; return; }
        if (in_op != SOP_NONE) push_feed<0>(in_op, in_len);
        in_op = op;
        in_len = len;
    }
    NPORE_STD_HD void finish()
    {
        if (in_op != SOP_NONE) push_feed<0>(in_op, in_len);
        in_op = SOP_NONE;
        push_end<0>();
        blk_end<0>();
        push_end<1>();
        blk_end<1>();
        e_end();
    }
};

}  // namespace npore
