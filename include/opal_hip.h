/*
 * opal_hip.h -- C-ABI of libopal_hip.so: the MI355X (gfx950) implementation of Opal's hot path
 * (fragment sampling -> LDPC spaced-k-mer features -> VW-style hashing -> one-against-all scoring).
 *
 * The reference (yunwilliamyu/opal) has no FFI: the path sits behind three in-process Python call sites and
 * one subprocess protocol.  Each entry point below names the reference interface it replaces (file:line under
 * the reference tree).  INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative OPAL_E* code;
 *     opal_last_error(ctx) (ctx may be NULL for handle-less calls) returns a NUL-terminated message.
 *   - pointers named d_* are DEVICE pointers (HBM) on the ctx's device; all others are host pointers.
 *   - device work is enqueued on the ctx's HIP stream and is asynchronous; opal_sync() waits for it.
 *   - one ctx per GPU, not thread-safe; different ctxs may be driven concurrently.
 *
 * Fragment batch layout in HBM ("reads"):
 *   d_frag_off[i]  where fragment i starts inside d_reads: a BYTE offset for raw fragments, a BASE offset (4 per byte)
 *                  for 2-bit packed ones, so that a packed fragment may alias any window of a packed genome that lives
 *                  in the same buffer (zero-copy fragment sampling: offset = 4*genome_byte_offset + position)
 *   d_frag_len[i]  low 31 bits: length in bases; bit 31 (OPAL_RAW_FLAG) set -> the fragment is stored as raw
 *                  ASCII bytes (one per base: anything that is not upper-case A/C/G/T lives here), clear -> it is
 *                  2-bit packed, 4 bases per byte, base j of the fragment in bits 2*(j&3) of byte j>>2, A=0 C=1 G=2 T=3.
 */
#ifndef OPAL_HIP_H
#define OPAL_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPAL_OK 0
#define OPAL_EINVAL (-22)   /* bad argument / shape                      */
#define OPAL_ENOMEM (-12)   /* host or device allocation failed          */
#define OPAL_ESTATE (-1)    /* patterns / weights not set yet            */
#define OPAL_EHIP (-5)      /* a HIP runtime call failed (see last error)*/
#define OPAL_ERANGE (-34)   /* output capacity too small                 */

#define OPAL_RAW_FLAG 0x80000000u
#define OPAL_MAX_K 256      /* k-mer length limit                        */
#define OPAL_MAX_T 64       /* row weight limit                          */
#define OPAL_MAX_P 64       /* number of patterns limit                  */
#define OPAL_VW_CONSTANT 11650396u

typedef struct opal_ctx opal_ctx;
typedef struct opal_rng opal_rng;

/* ---- library / context ------------------------------------------------------------------------------- */
const char *opal_version(void);
const char *opal_last_error(const opal_ctx *ctx);

/* stream: the hipStream_t every call of this ctx enqueues on (e.g. torch's current stream); NULL is HIP's
 * default (null) stream.  The ctx never creates or destroys streams.                                       */
int opal_ctx_create(int device_id, void *stream, opal_ctx **out);
int opal_ctx_destroy(opal_ctx *ctx);
int opal_sync(opal_ctx *ctx);
void *opal_ctx_stream(opal_ctx *ctx);

/* Replaces fasta2skm.create_pattern_getters (util/fasta2skm.py:120-134): pos is the P x t table of positions
 * read from patterns.txt (row-major, each 0 <= pos < k).  Needs k % t == 0 (fasta2skm.py:125).              */
int opal_set_patterns(opal_ctx *ctx, const int32_t *pos, int P, int t, int k);

/* Replaces `vw -t -i <model>` loading its regressor (opal.py:449-451).  w is the compact table of the
 * 2^bits `w` components (one float per VW weight slot), K the --oaa class count (opal.py:311).  The ctx keeps
 * a private device copy padded by K floats so that rows never wrap.  opal_load_weights takes a HOST table,
 * opal_set_weights_device a DEVICE table (copied device-to-device).  K is the number of distinct taxids
 * (opal.py:288) and may be any value in 1 .. 2^20: rows start at slot (h*K) mod 2^bits, i.e. at any slot when K is
 * odd, and are read with 16-byte loads on 4-byte alignment (no slow path for K that is not a multiple of 4).    */
int opal_load_weights(opal_ctx *ctx, const float *w, uint64_t n_slots, int K, int bits);
int opal_set_weights_device(opal_ctx *ctx, const float *d_w, uint64_t n_slots, int K, int bits);

/* Like opal_set_weights_device without the second copy: the ctx scores straight from the caller's buffer, which must
 * stay alive and unchanged while the ctx uses it and must hold capacity_floats >= n_slots + K + 4 floats, 16-byte
 * aligned.  The call writes the mirror of the table's first K+4 floats into [n_slots, n_slots+K+4) (rows then never
 * wrap); the ctx never frees or grows the buffer.  An 8 GiB model is resident once.  (opal.py:449-451, as above)     */
int opal_adopt_weights_device(opal_ctx *ctx, float *d_w, uint64_t n_slots, uint64_t capacity_floats, int K, int bits);

/* Tuning knobs for tests and experiments (the defaults are the production settings; the library reads no environment
 * variables).  predict_nt: -1 = row-load cache policy by table size (non-temporal from 4 GiB), 0 = default loads,
 * 1 = non-temporal loads.  emit_qchunk: fragments per work-queue draw of opal_features (at most 64), 0 = default (32).
 * emit_fast: 1 = opal_features hashes short 2-bit packed fragments with its fast kernel (default), 0 = generic kernel only;
 * 2 / 3 = a measured, not adopted experiment (per-base term arrays for the run pattern, fragments of up to 256 / 512 bases;
 * canonical two-pattern geometry only, same output); 4 = the fast kernels without the paired form of the gathered rows (two letters
 * per LDS word where the row's positions allow it: the default; tests compare the two).  train_group: examples per group of opal_train_batch's deterministic mode, 0 = default (8192).                        */
int opal_set_tuning(opal_ctx *ctx, int predict_nt, uint64_t emit_qchunk, int emit_fast, uint64_t train_group);

/* VW behaviours that differ between versions, defaults in brackets:
 * hash_all [0]: 0 = `--hash strings` (all-digit names are read as numbers), 1 = `--hash all`;
 * add_constant [1]: append the bias feature 11650396 after the user features;
 * clip [1]: clamp every class score to [min_label,max_label] (NaN -> 0) before the argmax [-1, 1].          */
int opal_set_vw_options(opal_ctx *ctx, int hash_all, int add_constant, int clip, float min_label, float max_label);

/* ---- host-side helpers (no GPU needed) ------------------------------------------------------------- */

/* Packs n sequences (concatenated ASCII in `seqs`, seq_off[n+1]) into the fragment batch layout above.
 * Call with out==NULL to learn the byte size (*out_bytes); frag_off[n] and frag_len[n] are always written
 * when non-NULL.  Replaces the text hand-off fasta_reader -> gen_features (util/fasta2skm.py:77-78).        */
int opal_pack_reads(const uint8_t *seqs, const uint64_t *seq_off, uint64_t n, uint8_t *out, uint64_t out_cap,
                    uint64_t *frag_off, uint32_t *frag_len, uint64_t *out_bytes);

/* CPython-exact `random` stream (MT19937), the generator drawfrag.py:24,38 and opal.py:363 draw from.       */
int opal_rng_create(opal_rng **out);
int opal_rng_destroy(opal_rng *rng);
/* random.seed(int): limbs = 32-bit little-endian limbs of abs(seed), n_limbs >= 1.                          */
int opal_rng_seed(opal_rng *rng, const uint32_t *limbs, int n_limbs);
/* random.randint(0, n-1) -> *out                                                                          */
int opal_rng_randbelow(opal_rng *rng, uint64_t n, uint64_t *out);
/* Replaces drawfrag.main_not_commandline's sampling loop (util/drawfrag.py:27-49, --atgc off as opal.py
 * always calls it): for each genome in order draws fragments until coverage*len is covered.  Writes at most
 * cap (position, genome index) pairs; *n_out is the number drawn (OPAL_ERANGE if it exceeds cap; the stream
 * is consumed either way, re-seed before retrying).                                                       */
int opal_sample_positions(opal_rng *rng, const uint64_t *genome_len, uint64_t n_genomes, uint64_t frag_len,
                          double coverage, uint64_t *out_pos, uint32_t *out_genome, uint64_t cap, uint64_t *n_out);
/* random.shuffle of [0..n) continuing the same stream (opal.py:363): perm must hold 0..n-1 on entry.       */
int opal_rng_shuffle(opal_rng *rng, uint64_t *perm, uint64_t n);

/* ---- device entry points ----------------------------------------------------------------------------- */

/* Per-fragment feature counts as an exclusive prefix sum: d_out_off[i] = sum_{j<i} F_j, d_out_off[n] = total,
 * F_j = max(0,len_j-k+1) * P * (reverse?2:1)  (get_all_substrings, util/fasta_functions.py:53-54).          */
int opal_feature_offsets(opal_ctx *ctx, const uint32_t *d_frag_len, uint64_t n, int reverse, uint64_t *d_out_off);

/* opal_feature_offsets for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for code under HIP
 * graph capture, tight serving loops and harnesses that forbid hipMalloc.  Same arguments and results; the per-tile
 * totals of the scan live in a static 512 KiB device array of the library (one copy per GPU, held until the process
 * ends) instead of the ctx's grown-on-demand buffer, so the call makes no hipMalloc / hipFree and does not synchronise.
 * Limit: n <= OPAL_NOALLOC_MAX_FRAGMENTS (OPAL_ERANGE beyond).  Every call on one GPU shares that GPU's array: calls
 * that overlap on one GPU (two ctxs / streams on the same device) race; calls on one stream are ordered and safe.   */
#define OPAL_NOALLOC_MAX_FRAGMENTS 134217728ull
int opal_feature_offsets_noalloc(opal_ctx *ctx, const uint32_t *d_frag_len, uint64_t n, int reverse, uint64_t *d_out_off);

/* Zero-copy drawfrag.  Replaces `sample = seq[pos:pos+size]` + writing the fragment FASTA (util/drawfrag.py:38-46) for
 * genomes that are resident in HBM as a fragment batch (d_genome_off / d_genome_len = that batch's d_frag_off /
 * d_frag_len): fragment i becomes a view of frag_len bases at position d_pos[i] of genome d_genome_idx[i] (the two
 * arrays opal_sample_positions fills), in the units of that genome (bases if packed, bytes if raw).  A draw that does
 * not fit inside its genome gets length 0.  The result is used with the genomes' own d_reads / reads_bytes.      */
int opal_fragment_views(opal_ctx *ctx, const uint64_t *d_genome_off, const uint32_t *d_genome_len, uint64_t n_genomes,
                        const uint64_t *d_pos, const uint32_t *d_genome_idx, uint64_t n, uint32_t frag_len,
                        uint64_t *d_frag_off, uint32_t *d_frag_len);

/* Emit mode. Replaces fasta2skm.gen_features + VW's name hashing for a whole batch
 * (util/fasta2skm.py:137-141,78-80; vw parse of the line written at opal.py:366-367,468-469):
 * d_out_idx[d_out_off[i] + f] = hash(name of feature f of fragment i) & (2^bits - 1), features in the
 * reference's order (forward strand window-major / pattern-minor, then the reverse complement).
 * d_out_off must have been filled by opal_feature_offsets with the same `reverse`.  out_cap = number of
 * indices d_out_idx can hold (normally d_out_off[n]); a fragment whose range would pass it is not written,
 * so offsets that belong to another batch cannot make the kernel write outside the buffer.  A fragment whose
 * (offset, length) reaches outside d_reads has no bases to hash but still owns a range (the offsets come from the
 * lengths alone): its range is filled with 0xffffffff (NUL bytes in opal_feature_names) and the next opal_sync
 * returns OPAL_EINVAL saying how many there were.                                                            */
int opal_features(opal_ctx *ctx, const uint8_t *d_reads, uint64_t reads_bytes, const uint64_t *d_frag_off,
                  const uint32_t *d_frag_len, uint64_t n, int reverse, int bits, const uint64_t *d_out_off,
                  uint32_t *d_out_idx, uint64_t out_cap);

/* Debug twin of opal_features: writes the t gathered characters of every feature (the feature NAME without
 * its decimal pattern suffix) so that the host can rebuild the reference's VW text line byte for byte
 * (util/fasta2skm.py:81-82).  d_out_names holds t bytes per feature at (d_out_off[i]+f)*t; out_cap = number
 * of FEATURES it can hold (t bytes each), checked like opal_features' out_cap.                                */
int opal_feature_names(opal_ctx *ctx, const uint8_t *d_reads, uint64_t reads_bytes, const uint64_t *d_frag_off,
                       const uint32_t *d_frag_len, uint64_t n, int reverse, const uint64_t *d_out_off,
                       uint8_t *d_out_names, uint64_t out_cap);

/* Fused predict. Replaces fasta2skm.main_generator piped into `vw -t -i model -p preds` (opal.py:449-469):
 * d_out_class[i] = 1-based VW class of fragment i (what vw writes to the -p file), d_out_scores (optional,
 * may be NULL) receives the K per-class scores of every fragment, row-major [n][K].
 * Asynchronous on the ctx stream; calls on one ctx are ordered by that stream (they share the ctx's work-queue
 * counter), so use one ctx per concurrent stream.                                                             */
int opal_predict(opal_ctx *ctx, const uint8_t *d_reads, uint64_t reads_bytes, const uint64_t *d_frag_off,
                 const uint32_t *d_frag_len, uint64_t n, int reverse, int32_t *d_out_class, float *d_out_scores);

/* ---- trainer (replaces the `vw --oaa K --bit_precision b --l1 --l2 --save_resume` learner process that
 * opal.py:307-327 starts and opal.py:363-367 feeds).  VW-like rule (adaptive AdaGrad
 * accumulators + importance-invariant squared-loss step, labels +1/-1 per class), not VW's exact arithmetic:
 * see opal_amd/csrc/opal_train.hip.  [VW-unpinned]                                                          */

/* Allocates (resume=0: zeroed) the 2^bits weight table and its AdaGrad accumulators.  resume=1 keeps the weights
 * already loaded in the ctx (same K and bits), like continuing from a --save_resume model; the accumulators are kept
 * too when opal_load_accumulators_sparse installed them (or an earlier training left them), else they restart at zero. */
int opal_train_begin(opal_ctx *ctx, int K, int bits, int resume);

/* One pass over a batch of labelled fragments, in the order given (the caller shuffles: opal.py:363).
 * d_labels[i] = VW class id (1..K) of fragment i (the "<label> |" prefix of the line, util/fasta2skm.py:82).
 * eta: learning rate (vw -l, default 0.5); l1/l2: opal.py --lambda1/--lambda2.  `sequential` selects the schedule:
 * 0 = all waves concurrently (Hogwild: fastest, results differ slightly from run to run);
 * 1 = the examples one after the other on a single wave (the sequential rule itself; slow: tests);
 * 2 = deterministic and concurrent: groups of examples (opal_set_tuning train_group) score against the table as it was
 *     when their group began, their updates are applied in one defined order by slot-range owners (opal_dtrain.hip):
 *     two runs give bit-identical tables.  Allocates its buffers on first use; synchronises the stream once per call.
 * d_out_pred (optional): the class predicted for each example BEFORE its update (progressive validation).   */
int opal_train_batch(opal_ctx *ctx, const uint8_t *d_reads, uint64_t reads_bytes, const uint64_t *d_frag_off,
                     const uint32_t *d_frag_len, const int32_t *d_labels, uint64_t n, int reverse, float eta, float l1,
                     float l2, int sequential, int32_t *d_out_pred);

/* Copies the 2^bits compact weight table out of the ctx (dst on the device if dst_is_device, else on the host):
 * what `vw -f model` would persist (opal.py:309).  Synchronises the stream.                                   */
int opal_get_weights(opal_ctx *ctx, float *dst, int dst_is_device);

/* Loads a model stored as nnz (slot index, weight) pairs (host arrays); all other slots are 0.                */
int opal_load_weights_sparse(opal_ctx *ctx, const uint32_t *idx, const float *val, uint64_t nnz, int K, int bits);

/* The learner's resume state: what `--save_resume` adds to a vw regressor (opal.py:310) is, for this rule, the AdaGrad
 * accumulator G of every slot.  opal_get_accumulators copies the 2^bits accumulators out (after opal_train_begin /
 * opal_train_batch); opal_load_accumulators_sparse installs nnz (slot, G) pairs (all other slots 0) next to weights that are
 * already loaded with the same K and bits, so that opal_train_begin(resume=1) continues the rule where the saved run
 * stopped instead of restarting the accumulators at zero.  [VW-unpinned: the reference has no resume path of its own]   */
int opal_get_accumulators(opal_ctx *ctx, float *dst, int dst_is_device);
int opal_load_accumulators_sparse(opal_ctx *ctx, const uint32_t *idx, const float *val, uint64_t nnz);

/* GPU-side FASTA ingest.  Replaces fasta_functions.fasta_reader over open(path,'r')
 * (util/fasta_functions.py:9-43, universal newlines, '+' mutes up to the next '>', text before the first '>' is
 * dropped, an empty last record is dropped) followed by packing, for a whole file whose bytes are already in HBM:
 * d_text[nbytes] -> fragment batch (d_reads_out, d_frag_off, d_frag_len), one fragment per record in record order.
 * All sequence bytes of the file go into ONE 2-bit stream at the start of d_reads_out; a record made of upper-case
 * ACGT is a view into it (d_frag_off in bases, not byte-aligned), any other record is stored as raw bytes behind the
 * stream (in no particular order).  d_reads_out: 4-byte aligned, reads_cap >= nbytes + nbytes/4 + 64 always suffices;
 * d_frag_off/d_frag_len: max_records entries (nbytes/2 + 1 always suffices).  The text is read once (a single-pass
 * chained scan), so the call needs no sizes in advance: with a buffer that turns out too small it returns OPAL_ERANGE
 * and reports, through the same two host pointers, the record count and the d_reads_out bytes a second call needs
 * (bytes = 0, not known yet, when it was max_records that was too small).
 * Record names are not produced.  Synchronises the stream and returns the record count and the bytes of d_reads_out
 * in use through the two host pointers.                                                                        */
int opal_ingest_fasta(opal_ctx *ctx, const uint8_t *d_text, uint64_t nbytes, uint8_t *d_reads_out, uint64_t reads_cap,
                      uint64_t *d_frag_off, uint32_t *d_frag_len, uint64_t max_records, uint64_t *n_records_out,
                      uint64_t *reads_bytes_out);

/* Sizes only: the number of '>' header lines (the record count, or one less when the last record is empty; a text
 * without any header is one record) and of sequence bytes in the text, for callers that want exact buffers before
 * opal_ingest_fasta: max_records = max(headers, 1); d_reads_out = seq_bytes/4 + 64 if every record is upper-case ACGT,
 * plus the bytes of the records that are not (at most seq_bytes more).  One more pass over the text (tile summaries +
 * a two-level scan); opal_ingest_fasta does not need it.  Synchronises the stream.                               */
int opal_ingest_scan(opal_ctx *ctx, const uint8_t *d_text, uint64_t nbytes, uint64_t *n_headers_out, uint64_t *seq_bytes_out);

/* Evaluation counts.  Replaces the string compare + groupby of opal.evaluate_predictions (opal.py:111-131):
 * d_total[c] = number of fragments whose reference class is c, d_correct[c] = those predicted as c.
 * d_pred/d_ref: dense class ids in [0, n_classes) (ref) / any int (pred); both outputs are zeroed by the call.
 * micro = sum(correct)/n, per-class accuracy = correct/total -> macro (mean), median.                         */
int opal_eval_counts(opal_ctx *ctx, const int32_t *d_pred, const int32_t *d_ref, uint64_t n, uint32_t n_classes,
                     uint64_t *d_correct, uint64_t *d_total);

/* Name of the scoring-kernel variant opal_predict would launch for the current K (for profiles/bench).   */
const char *opal_predict_kernel_name(opal_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* OPAL_HIP_H */
