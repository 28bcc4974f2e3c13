/*
 * npore_amd.h -- C ABI of libnpore_amd.so: the MI355X (gfx950) implementation of
 * nPoRe's per-read realignment DP.
 *
 * The reference has no FFI layer; its boundary for this path is the Cython
 * extension module `aln` (reference src/aln.pyx), star-imported by its callers
 * (src/realign.py:11, src/bam.pyx:12).  The entry points below are what a
 * ctypes binding replacing that module needs; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers and sizes only; all buffers are caller-owned;
 *   - base codes: N=0 A=1 C=2 G=3 T=4 (reference src/cfg.py:11-26);
 *   - a CIGAR is the *expanded* op string over "=XMID" with S/H already
 *     stripped (what src/bam.pyx:59 hands to align());
 *   - every function returns 0 on success or a negative NPORE_E_* code and
 *     leaves a message retrievable with npore_last_error() (thread local).
 *     The reference prints "ERROR ..." and exit(1)s or silently truncates
 *     (src/aln.pyx:689-716); here the same conditions are reported per read
 *     in `status` (NPORE_ST_* bits) and the truncated string is still returned.
 *   - one context per GPU; a context is not re-entrant (one call at a time; asynchronous batches overlap
 *     on the device, not in the API).
 */
#ifndef NPORE_AMD_H
#define NPORE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPORE_ABI_VERSION 2

/* return codes */
#define NPORE_OK 0
#define NPORE_E_INVALID (-1)   /* bad argument */
#define NPORE_E_NODEVICE (-2)  /* no usable HIP device / wrong architecture */
#define NPORE_E_HIP (-3)       /* HIP runtime error (see npore_last_error) */
#define NPORE_E_NOMEM (-4)
#define NPORE_E_UNSUPPORTED (-5) /* parameter outside what the kernels cover */

/* per-read status bits (0 = clean) */
#define NPORE_ST_ROW_NEG 1      /* reference: "ERROR: row < 0"   src/aln.pyx:689 */
#define NPORE_ST_COL_NEG 2      /* reference: "ERROR: col < 0"   src/aln.pyx:699 */
#define NPORE_ST_RUN_LT1 4      /* reference: "ERROR: run 0"     src/aln.pyx:708 */
#define NPORE_ST_BAD_TYPE 8     /* reference: unknown matrix type src/aln.pyx:737 */
#define NPORE_ST_OUT_OF_CHUNK 16 /* traceback left the chunk (UB in the reference) */
#define NPORE_ST_BAD_INPUT 32   /* CIGAR/sequence lengths disagree, bad op or base code */
#define NPORE_ST_OUT_CAP 64     /* caller's output slot too small */

typedef struct npore_ctx npore_ctx;

/* ABI version of the loaded library (== NPORE_ABI_VERSION it was built with). */
int npore_abi_version(void);

/* Message for the last failure on this thread ("" if none). */
const char *npore_last_error(void);

/* Number of visible gfx950 devices (0 if none / HIP unavailable). */
int npore_device_count(void);

/*
 * Create a context on HIP device `device_id` holding the penalty tables.
 *   sub_scores f32[5][5], np_scores f32[max_n][max_l+1][max_l+1]: the outputs of
 *   calc_score_matrices (reference src/aln.pyx:62-96), which align() receives as
 *   arguments (src/aln.pyx:380).  max_n/max_l replace the reads of
 *   cfg.args.max_n / cfg.args.max_l (src/aln.pyx:436-437).  Limits: 1 <= max_n <= 6, 2 <= max_l <= 127.
 *   sub_scores == np_scores == NULL makes an annotation-only context: npore_get_np_info and
 *   npore_np_regions work on it (the reference's get_np_info takes no tables, src/aln.pyx:179;
 *   its callers src/bed.py:62 and src/bam.pyx:381 never call align() first), the align entry
 *   points return NPORE_E_INVALID.
 * Returns NULL on failure.
 */
npore_ctx *npore_ctx_create(const float *sub_scores, const float *np_scores,
                            int max_n, int max_l, int device_id);
void npore_ctx_destroy(npore_ctx *ctx);

/*
 * Batched replacement for align() (reference src/aln.pyx:379-787); one call of
 * the reference == a batch of one.  Host buffers in, host buffers out.
 *   refs/seqs/cigars: concatenated byte buffers; *_off: int64[n_reads+1] prefix
 *   offsets.  out: caller buffer; read i may use out[out_off[i] .. out_off[i+1]);
 *   its length goes to out_len[i] (an alignment never exceeds |ref|+|seq| ops).
 *   status[i]: NPORE_ST_* bits.
 * A batch is worked through in groups of reads (as many as the traceback budget holds); every group uploads its
 * slice of the inputs and downloads its slice of the results around its own kernels, so that the copies of one
 * group run beside the kernels of its neighbours.
 * Results are bit-identical to the reference for the same
 * (read, r, max_b_rows, indel_start, indel_extend, max_n, max_l, tables).
 */
int npore_align_batch(npore_ctx *ctx, int64_t n_reads,
                      const uint8_t *refs, const int64_t *ref_off,
                      const uint8_t *seqs, const int64_t *seq_off,
                      const char *cigars, const int64_t *cig_off,
                      float indel_start, float indel_extend,
                      int max_b_rows, int r,
                      char *out, const int64_t *out_off,
                      int64_t *out_len, int32_t *status);

/*
 * The same, returning once the batch is enqueued: up to two groups of reads of a context are on the device at a
 * time (uploads, kernels and downloads of neighbouring groups overlap), and the results are complete when
 * npore_ctx_wait() returns.  ALL arrays passed in -- inputs, offsets and outputs -- must stay valid and untouched
 * until then; page-locked host memory (hipHostMalloc / a pinned torch tensor) makes the copies truly asynchronous,
 * pageable memory works but serialises them.  A failure found later is reported by the next call on the context.
 */
int npore_align_batch_async(npore_ctx *ctx, int64_t n_reads,
                            const uint8_t *refs, const int64_t *ref_off,
                            const uint8_t *seqs, const int64_t *seq_off,
                            const char *cigars, const int64_t *cig_off,
                            float indel_start, float indel_extend,
                            int max_b_rows, int r,
                            char *out, const int64_t *out_off,
                            int64_t *out_len, int32_t *status);

/*
 * align() AND realign_read's glue for a batch (reference src/bam.pyx:59-78: align(), then X,= -> M, one pass of
 * push-D-left / I-through-D / push-I-left / I-through-D (src/cig.pyx:102-192), 'ID' -> 'M', collapse_cigar
 * src/cig.pyx:13-38), all on the device: out receives the collapsed, standardised CIGAR TEXT of read i in
 * out[out_off[i] .. out_off[i] + out_len[i]) instead of the op string; slots of 2 * (ref + read bases) + 16 bytes always
 * suffice (NPORE_ST_OUT_CAP otherwise).  A read align() refuses gets an empty text.  Same bytes as
 * npore_align_batch followed by npore_standardize_batch; the op strings never leave the GPU.  Synchronous.
 */
int npore_align_batch_cigars(npore_ctx *ctx, int64_t n_reads,
                            const uint8_t *refs, const int64_t *ref_off,
                            const uint8_t *seqs, const int64_t *seq_off,
                            const char *cigars, const int64_t *cig_off,
                            float indel_start, float indel_extend,
                            int max_b_rows, int r,
                            char *out, const int64_t *out_off,
                            int64_t *out_len, int32_t *status);

/*
 * Device-resident variant used by bench.py and by pipelines that already hold
 * the reads in HBM: same arguments, but every pointer except ctx is a DEVICE
 * pointer (hipMalloc'ed by the caller, e.g. a torch.cuda tensor's data_ptr()).
 * `stream` is a hipStream_t passed as void*: the batch is ordered behind the work that stream holds at
 * the time of the call (NULL = no such order: the context's own streams are non-blocking, they are NOT
 * ordered with work on the NULL stream, so that several contexts run side by side -- the buffers must be
 * ready when the call is made).
 * sync != 0: the call returns after the batch has completed.
 * sync == 0: the call returns once the batch is enqueued; up to two batches of a context are in flight, a
 *   third call first waits for the oldest.  A context keeps two sets of work buffers and three streams
 *   (preparation | fill kernel | traceback + gather), so the next batch is prepared and the previous one traced
 *   back while the fill kernel works on the current one.  Results are complete when npore_ctx_wait() returns
 *   (and for work put on `stream` after the call, if one was passed); the input and output buffers must stay
 *   untouched until then.  A failure found later is reported by the next call on the context.
 * A batch too large for the traceback budget is split into groups that go through the same pipeline inside
 * one call, whatever `sync` says.
 */
int npore_align_batch_device(npore_ctx *ctx, int64_t n_reads,
                             const uint8_t *d_refs, const int64_t *d_ref_off,
                             const uint8_t *d_seqs, const int64_t *d_seq_off,
                             const char *d_cigars, const int64_t *d_cig_off,
                             float indel_start, float indel_extend,
                             int max_b_rows, int r,
                             char *d_out, const int64_t *d_out_off,
                             int64_t *d_out_len, int32_t *d_status,
                             void *stream, int sync);

/* Wait for every batch enqueued with sync == 0; returns the first failure among them. */
int npore_ctx_wait(npore_ctx *ctx);

/*
 * get_np_info (reference src/aln.pyx:179-251): n-polymer annotation of one
 * sequence (len < 2^30).  out is int32[len][2][max_n] ([pos][0=L,1=L_IDX][n-1]).
 * One kernel launch per period over as many workgroups as the sequence has
 * 64-base windows; work buffers are kept in the context.
 */
int npore_get_np_info(npore_ctx *ctx, const uint8_t *seq, int64_t len, int32_t *out);

/*
 * get_np_regions (reference src/bed.py:56-76) for a batch of independent slices of a genome (the reference
 * cuts every region into --chunk_width pieces, src/bam.pyx:149-162, and annotates each on its own):
 * get_np_info on every slice, then the n-polymer starts (L != 0 and L_IDX == 0) per period.
 * seqs: base codes, slice k = seqs[seq_off[k] .. seq_off[k+1]).  counts[(n-1) * n_slices + k] = starts of
 * period n in slice k; *pos / *reps (position within the slice, repeat count L) list them ordered by
 * (period, slice, position) -- the region of an entry is [pos, pos + n * L).  *pos and *reps point into
 * storage owned by ctx, valid until the next call on it.
 */
int npore_np_regions(npore_ctx *ctx, const uint8_t *seqs, const int64_t *seq_off, int64_t n_slices,
                     int64_t *counts, const int32_t **pos, const int32_t **reps, int64_t *total);

/*
 * Host-side glue after align(): what realign_read does with the returned string
 * (reference src/bam.pyx:65-78 with src/cig.pyx:102-192 and collapse_cigar, src/cig.pyx:13-38):
 * X,= -> M, one pass of push-D-left / I-through-D / push-I-left / I-through-D, 'ID' -> 'M',
 * run-length encoding.  alns: concatenated align() strings; out receives the collapsed
 * CIGAR text of read i in out[out_off[i] .. out_off[i+1]) (2 bytes per op is always enough).
 * No GPU involved; `threads` <= 0 means all host cores.
 */
int npore_standardize_batch(int64_t n_reads, const char *alns, const int64_t *aln_off,
                            const uint8_t *refs, const int64_t *ref_off,
                            const uint8_t *seqs, const int64_t *seq_off,
                            char *out, const int64_t *out_off, int64_t *out_len, int threads);
/* The same without the run-length encoding: the expanded op string over 'MID' that realign_hap returns
 * (reference src/bam.pyx:116); read i needs at most as many bytes as its align() string has. */
int npore_standardize_ops_batch(int64_t n_reads, const char *alns, const int64_t *aln_off,
                                const uint8_t *refs, const int64_t *ref_off,
                                const uint8_t *seqs, const int64_t *seq_off,
                                char *out, const int64_t *out_off, int64_t *out_len, int threads);

/*
 * Timing of the stages of the last npore_align_batch* call on this context,
 * measured with HIP events on the stream the kernels ran on (milliseconds):
 *   ms[0] device prep kernels, ms[1] fill kernel(s), ms[2] traceback + gather,
 *   ms[3] H2D, ms[4] D2H (host-buffer entry points only; summed over the groups), ms[5] unused (0),
 *   ms[6] cells processed (count), ms[7] fill-kernel launches.
 */
int npore_last_timing(npore_ctx *ctx, double *ms, int n);
/* The same entries summed over every group of reads the context has completed since it was made (for a caller
 * that keeps batches in flight: differences over a timed region). */
int npore_total_timing(npore_ctx *ctx, double *ms, int n);

/* Tunables: key in {"tb_budget_mb","force_chunks","coresident"} (traceback budget in MiB per
 * work set, chunks per fill workgroup).  "coresident"
 * (default 1): a group of reads that overlaps another one on the device is prepared and gathered by kernel shapes
 * that run beside the fill kernel's workgroups, and the fill leaves them room; 0 = always the stand-alone shapes).
 * "fill_streams" (1 | 2, default 2): fill kernels of consecutive groups on one stream or alternating between two.
 * The BAM -> SAM pipeline (npore_bam_realign_file / _sequential): "device_glue" (default 1): realign_read's glue
 * (src/bam.pyx:65-78) on the device, 0 = on the host from the op strings; "device_pack" (default 1, with the device
 * glue): align()'s inputs (src/bam.pyx:42, 45, 59-61) unpacked from the BAM records on the device, 0 = packed on the
 * host and uploaded. */
int npore_ctx_set(npore_ctx *ctx, const char *key, int64_t value);

/* Batch sizing.  The DP of a chunk (at most max_b_rows anti-diagonals of a read; reference src/aln.pyx:344-358,
 * 445-456) is one unit of work of the fill kernel: a chain of dependent anti-diagonals that a group of wavefronts
 * works through (~1.1 us each).  The GPU holds npore_round_chunks(ctx, r) such groups at a time -- 4 096 at r = 30,
 * 1 024 at r = 100, 512 at r = 200 on an MI355X -- and every group pulls its next chunk from a queue (largest first)
 * the moment it has finished one, so a batch costs about ceil(full-size chunks / resident groups) chain times: at
 * r = 30, 4 000 reads of 10 kb take 21.7 ms, 4 500 take 30.6 ms (not two full rounds), and from 6 000 reads on the
 * rate stays within ~12 % of its maximum (profiles/r02_sweep_r30_queue.csv).  Batches enqueued with sync = 0 follow
 * each other without a gap.  Returns the resident groups, 0 for a band the kernels do not cover (r > 511).
 * Replaces nothing in the reference (its pool has no such granularity, src/realign.py:110-114). */
int64_t npore_round_chunks(npore_ctx *ctx, int r);
/* Launch geometry of the fill kernel at band half-width r, for reports (bench.py's "practical bound"):
 * out[0] wavefronts per chunk, [1] chunks per workgroup, [2] workgroups per CU, [3] workgroups resident on the
 * GPU, [4] LDS bytes per workgroup (up to n entries are written). */
int npore_fill_shape(npore_ctx *ctx, int r, int32_t *out, int n);

/* ---------------------------------------------------------------------------------------------
 * BAM ingest / SAM emit around the batched align(): the host side of realign_read
 * (reference src/bam.pyx:18-47 get_read_data, :51-84 realign_read, with pysam's fetch and
 * per-read accessors underneath), native and for a whole batch of reads at a time.
 * npore_amd/bam.py holds a pure-Python restatement of the same logic that the tests compare with.
 * No GPU involved except in npore_bam_realign_batch.  `threads` <= 0 means all host cores.
 */
typedef struct npore_bam npore_bam;
typedef struct npore_fasta npore_fasta;

/* Inflate (BGZF blocks in parallel) and index a BAM file; NULL on failure (reference: "ERROR: BAM file
 * ... not found", src/bam.pyx:22-24). */
npore_bam *npore_bam_open(const char *path, int threads);
/* The same with the ingest mode chosen: 0 = automatic (what npore_bam_open does: STREAMED when the file is BGZF and
 * larger than NPORE_BAM_STREAM_MB, default 1024), 1 = the whole inflated stream resident, 2 = streamed.
 * A STREAMED handle never holds the inflated stream (reference src/bam.pyx:18-47 iterates region by region through
 * pysam; its memory is O(one read)): it keeps the BGZF block table and 22 bytes per record (offset, reference,
 * position, span, flag -- what npore_bam_select needs), built in one pass over windows of 256 MB, and every batch
 * (npore_bam_pack*, _format_sam, _realign_batch, _realign_file) inflates just the blocks its records lie in.
 * index_path (may be NULL): a record index written by npore_bam_save_index for this file; the handle then skips its
 * indexing pass -- with one process per GPU, one process of a node indexes and the others load.
 * mode 3 (ONE-PASS): only the BGZF block table and the BAM header are read; the records are visited once, in file
 * order, by npore_bam_realign_sequential (such a handle has no record index: npore_bam_select finds nothing). */
npore_bam *npore_bam_open_mode(const char *path, int threads, int mode, const char *index_path);
int npore_bam_is_streamed(const npore_bam *bam);
int npore_bam_save_index(const npore_bam *bam, const char *path);
void npore_bam_close(npore_bam *bam);
/* Write the inflated BAM stream to `path` (complete or not at all: temporary file + rename).  npore_bam_open
 * recognises such a file ("BAM\1" at offset 0) and maps it instead of inflating: with several processes per node
 * (one per GPU) one of them inflates and the others share its copy through the page cache. */
int npore_bam_dump_inflated(const npore_bam *bam, const char *path);
int64_t npore_bam_inflated_size(const npore_bam *bam);
int64_t npore_bam_n_records(const npore_bam *bam);
int npore_bam_n_refs(const npore_bam *bam);
const char *npore_bam_ref_name(const npore_bam *bam, int i);
int64_t npore_bam_ref_len(const npore_bam *bam, int i);
int npore_bam_ref_has_reads(const npore_bam *bam, int i);

/* The reads get_read_data yields (src/bam.pyx:26-33): for each region (BAM reference id, [start, stop)) the
 * records overlapping it in file order, stopping after max_reads (0 = no cap), skipping secondary,
 * supplementary and unmapped ones.  Writes up to `cap` record indices; returns the total count (or < 0). */
int64_t npore_bam_select(const npore_bam *bam, int n_regions, const int32_t *ref_id, const int64_t *start,
                         const int64_t *stop, int64_t max_reads, int64_t *out_idx, int64_t cap);

/* {contig: upper-cased sequence} of a FASTA file (reference: pysam.FastaFile / Bio.SeqIO callers). */
npore_fasta *npore_fasta_open(const char *path);
void npore_fasta_close(npore_fasta *fa);
int npore_fasta_n(const npore_fasta *fa);
const char *npore_fasta_name(const npore_fasta *fa, int i);
int64_t npore_fasta_len(const npore_fasta *fa, int i);
/* the upper-cased bases of contig i (npore_fasta_len(fa, i) bytes, not NUL-terminated), owned by fa */
const char *npore_fasta_seq(const npore_fasta *fa, int i);

/* Inputs of npore_align_batch for the selected records (src/bam.pyx:59-61): expanded CIGAR without S/H,
 * query bases without the soft clips and the reference slice [pos, pos + reference_length) as codes.
 * fasta_of_ref[bam reference id] = index of that contig in `fa` (or -1).  Sizes first, then the fill. */
int npore_bam_pack_sizes(const npore_bam *bam, const int64_t *idx, int64_t n, int64_t *ref_off, int64_t *seq_off,
                         int64_t *cig_off);
int npore_bam_pack(const npore_bam *bam, const npore_fasta *fa, const int32_t *fasta_of_ref, const int64_t *idx,
                   int64_t n, uint8_t *refs, const int64_t *ref_off, uint8_t *seqs, const int64_t *seq_off,
                   char *cigs, const int64_t *cig_off, int threads);

/* SAM records (src/bam.pyx:83) of the selected reads given their final collapsed CIGARs
 * (finals[final_off[k] .. +final_len[k])); reads whose status has NPORE_ST_BAD_INPUT are left out.
 * *sam points into storage owned by `bam`, valid until the next call on it. */
int npore_bam_format_sam(npore_bam *bam, const int64_t *idx, int64_t n, const char *finals, const int64_t *final_off,
                         const int64_t *final_len, const int32_t *status, int threads, const char **sam,
                         int64_t *sam_len);

/* realign_read for a batch: pack -> npore_align_batch -> npore_standardize_batch -> SAM text. */
int npore_bam_realign_batch(npore_ctx *ctx, npore_bam *bam, const npore_fasta *fa, const int32_t *fasta_of_ref,
                            const int64_t *idx, int64_t n, float indel_start, float indel_extend, int max_b_rows,
                            int r, int threads, const char **sam, int64_t *sam_len, int32_t *status);
/* The same for all `n` selected reads in batches of `batch_reads`, appended to the file `out_path` in input
 * order, with the three stages overlapped: while the GPU aligns batch k, batch k+1 is packed and batch k-1
 * standardised, formatted and written.  status[n] as for npore_align_batch. */
int npore_bam_realign_file(npore_ctx *ctx, npore_bam *bam, const npore_fasta *fa, const int32_t *fasta_of_ref,
                           const int64_t *idx, int64_t n, int64_t batch_reads, float indel_start, float indel_extend,
                           int max_b_rows, int r, int threads, const char *out_path, int32_t *status);
/* ONE-PASS ingest: the whole BAM -> realigned SAM run in one sequential pass over the file (reference src/realign.py:
 * 100-115 with src/bam.pyx:18-47: get_read_data is a generator over bam.fetch(), read by read, feeding the pool).  The
 * stream is inflated ONCE, window by window (64 MB), the records are filtered as they go by (at most one region per
 * contig, in the order of the BAM header: the tool's default whole-contig regions; same overlap / flag / max_reads
 * rules as npore_bam_select), batched, and sent through the same overlapped stages as npore_bam_realign_file -- no
 * record index, no second inflation, memory bounded by the batches in flight.  `bam`: a handle opened in mode 3
 * (header only) or 2.  counts[3]: reads selected, reads refused (NPORE_ST_BAD_INPUT, not written), reads with an
 * inconsistent traceback; bad_ord / bad_status[bad_cap]: ordinal (among the selected reads) and status bits of the
 * first such reads.  NPORE_E_UNSUPPORTED: the regions or the file's order rule the one-pass run out (the BAM is not
 * sorted by reference, several regions per contig) -- nothing usable was written: truncate and use the indexed path. */
/* Several processes (one per GPU) on ONE file, each in one pass (the reference feeds all its workers from one sequential
 * read, src/bam.pyx:18-47 + src/realign.py:110-114): process `rank` of `world` walks a contiguous stretch of the record
 * stream -- the records that start between two cut points taken from the LINEAR index of the file's .bai (bai_path; each
 * entry is the virtual offset of a record: SAM specification 5.2), chosen at rank / world of the compressed file -- so every
 * block is inflated once per node, no record index is built, and the ranks' part files concatenated in rank order are in
 * file order.  The next npore_bam_realign_sequential on the handle walks that stretch only (max_reads must be 0).
 * world == 1 clears the share.  NPORE_E_UNSUPPORTED: no usable .bai (the caller takes the indexed reader). */
int npore_bam_set_share(npore_bam *bam, int rank, int world, const char *bai_path);
/* out4: [0] a share is set, [1] / [2] its first byte and its end in the inflated stream (-1: nothing / the end of the
 * file), [3] the BGZF block it begins in (tests). */
int npore_bam_share_info(const npore_bam *bam, int64_t *out4);
int npore_bam_realign_sequential(npore_ctx *ctx, npore_bam *bam, const npore_fasta *fa, const int32_t *fasta_of_ref,
                                 int n_regions, const int32_t *ref_id, const int64_t *start, const int64_t *stop,
                                 int64_t max_reads, int64_t batch_reads, float indel_start, float indel_extend,
                                 int max_b_rows, int r, int threads, const char *out_path, int64_t *counts,
                                 int64_t *bad_ord, int32_t *bad_status, int64_t bad_cap);
/* Host wall time of the stages of the last npore_bam_realign_batch on `bam` (milliseconds):
 * ms[0] pack, ms[1] npore_align_batch (incl. PCIe), ms[2] standardise, ms[3] SAM formatting. */
int npore_bam_last_timing(const npore_bam *bam, double *ms, int n);
/* Stage clocks of the last npore_bam_realign_file on `bam` (milliseconds): [0] record fetch + pack, [1] the align calls
 * (host wall inside npore_align_batch: upload, kernels, download), [2] standardise, [3] SAM text, [4] file write --
 * each the SUM over the batches of the time that stage's thread spent (the stages of neighbouring batches overlap);
 * [5] wall time of the call, [6] GPU kernel time (preparation + fill + traceback, both contexts), [7] PCIe time. */
int npore_bam_file_timing(const npore_bam *bam, double *ms, int n);

/*
 * The counting loop of calc_confusion_matrices (reference src/bam.pyx:351-499) for one range (ctg, start, end):
 * the basecaller's SUB / n-polymer / INS / DEL confusion counts from pileup text.  Host code, no GPU.
 *   lines / line_off[n_lines+1]: column 5 of `samtools mpileup -r ctg:start+1-end` (src/bam.pyx:301-316), upper-cased,
 *     one entry per reported position, taken as positions start, start+1, ... like the reference does;
 *   ref_codes[n_ref]: base codes of refs[ctg][start:end]; ref_text[ref_text_len]: the upper-cased contig from
 *     `start` to its end (the n-polymer unit an insertion is compared with, :459-460);
 *   np_info: get_np_info(refs[ctg][start:end+1]) as npore_get_np_info returns it, np_len positions.
 * The counts are ADDED to subs[5][5], nps[max_n][max_l+1][max_l+1], inss[max_l+1], dels[max_l+1] (int64), so that
 * ranges accumulate (the reference sums its pool's results, :183-188).  *bad_lines = lines holding a character the
 * reference reports as unexpected (:473-476; the rest of such a line is dropped there too). */
int npore_confusion_counts(const char *lines, const int64_t *line_off, int64_t n_lines, const uint8_t *ref_codes,
                           int64_t n_ref, const char *ref_text, int64_t ref_text_len, const int32_t *np_info,
                           int64_t np_len, int max_n, int max_l, int64_t *subs, int64_t *nps, int64_t *inss,
                           int64_t *dels, int64_t *bad_lines, int threads);

/* Debug / tests: one raw DEFLATE stream (a BGZF block's payload; the BAM reader's inner loop, pysam / htslib in the
 * reference) of in_len bytes that must inflate to exactly out_len bytes.  force: 0 = as the readers do it (this
 * library's decoder, csrc/inflate.hpp, and zlib for what it declines), 1 = the decoder only, 2 = zlib only.
 * Returns 1 (inflated), 0 (refused / malformed) or a negative NPORE_E_* code.  Host code, no GPU. */
int npore_debug_inflate(const uint8_t *in, int64_t in_len, uint8_t *out, int64_t out_len, int force);
/* The same for TWO independent streams decoded side by side by one thread, as the BGZF readers take a file's blocks
 * (csrc/inflate.hpp: two dependent chains in one loop).  Returns bit 0 = stream a inflated, bit 1 = stream b inflated,
 * or a negative NPORE_E_* code. */
int npore_debug_inflate_pair(const uint8_t *in_a, int64_t in_len_a, uint8_t *out_a, int64_t out_len_a, const uint8_t *in_b,
                             int64_t in_len_b, uint8_t *out_b, int64_t out_len_b, int force);
/* The CRC-32 (RFC 1952) the BGZF readers check every member's inflated bytes with (csrc/crc32.hpp: folding by carry-less
 * multiplication where the CPU has it): crc32(crc, p, n) of zlib, `crc` = the value so far (0 at the start).  Returns the
 * new value (0 ... 2^32 - 1) or a negative NPORE_E_* code.  Host code, no GPU. */
int64_t npore_debug_crc32(const uint8_t *p, int64_t n, uint32_t crc);

/* Debug self-test: out128[l] = value lane l receives from lane l-1 (l>0),
 * out128[64+l] = value from lane l+1 (l<63); checks the DPP wave-shift
 * directions the fill kernel relies on. */
int npore_debug_dpp(uint32_t *out128);

/* Debug self-test: exhaustive check of the kernels' small-divisor division
 * (run / n, 0 <= run < 65536, 1 <= n <= 6); *mismatches must come back 0. */
int npore_debug_divcheck(int64_t *mismatches);

/* Debug: copy a device-prepared array of the last group of the last align call
 * to the host (what: 0 steps, 1 inss, 2 chunk descriptors, 3 seqw, 4 refw,
 * 5 refl, 6 schedule, 7 counters); the GPU tests compare them with the host twin. */
int npore_debug_fetch(npore_ctx *ctx, int what, void *dst, int64_t bytes);

/* Debug: `bytes` of the last group's traceback words (u32 per cell: typ | run << 3,
 * src/aln.pyx:670-742 reads the same from its state matrix) from byte `offset` on;
 * a chunk's words start at 4 * tb_off of its descriptor. */
int npore_debug_fetch_tb(npore_ctx *ctx, int64_t offset, void *dst, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* NPORE_AMD_H */
