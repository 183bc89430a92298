/*
 * clairsto_amd.h -- C ABI of the MI355X (gfx950) hot-path engine for ClairS-TO.
 *
 * The hot path is: pileup-tensor creation -> AFF (CvT) + NEG (BiGRU)
 * inference -> posterior / arg-max / QUAL.  The reference has no FFI for this path; it sits behind
 * three concrete seams.  Every entry point below names the reference interface it
 * replaces (file:line into HKU-BAL/ClairS-TO v0.4.4).
 *
 * Conventions
 *   - plain C, no exceptions; every function returns CTO_OK (0) or a negative error code;
 *     cto_last_error() gives a thread-local message for the last failure.
 *   - "dev" pointers are device (HBM) pointers the caller owns; "host" pointers are host memory.
 *   - stream arguments are a hipStream_t passed as void* (NULL = the default stream).  All device
 *     work is stream-ordered; nothing synchronises the device unless stated.
 *   - handles are re-entrant per handle; one handle must not be used from two streams at once
 *     (its activation workspace is shared).
 */
#ifndef CLAIRSTO_AMD_H
#define CLAIRSTO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTO_OK            0
#define CTO_EINVAL       -1   /* bad argument / malformed input                        */
#define CTO_EHIP         -2   /* a HIP runtime call failed                              */
#define CTO_ENOMEM       -3
#define CTO_EUNSUPPORTED -4   /* input outside the documented limits    */
#define CTO_EMISSING     -5   /* a required weight tensor is missing / has a wrong size */

#define CTO_NPOS      33      /* shared/param.py:60-61  no_of_positions                 */
#define CTO_NCHAN     34      /* shared/param.py:50-58  pileup_channel_size             */
#define CTO_FLANK     16      /* shared/param.py:60     flankingBaseNum                 */

const char* cto_last_error(void);
int cto_version(void);
/* number of visible HIP devices, or a negative error code */
int cto_device_count(void);
/* Test aid: fills the LDS of every compute unit of the current device with signalling-NaN bit patterns (stream-ordered).
 * Kernels that read LDS they did not write (e.g. padding columns multiplied by zero weights) then fail parity tests
 * deterministically instead of depending on what the previous kernel left behind.  No reference counterpart. */
int cto_debug_poison_lds(void* stream);

/* ------------------------------------------------------------------------------------------------
 * Column pack: the binary form of `samtools mpileup` rows that the featurisation kernels consume.
 * One pack = one contig region, columns in increasing position order (one column per mpileup row).
 *
 * entry (uint32), one per read-base of a column, in samtools' read order, as printed with --min-BQ 0:
 *   bits  3:0  base code: 0..3 = A C G T, 4..7 = a c g t, 8 = '*', 9 = '#', 10 = 'N', 11 = 'n'
 *   bits  5:4  indel kind attached to this base: 0 none, 1 insertion, 2 deletion,
 *              3 = indel longer than max_indel_length (contributes nothing to the tensors, F4; still
 *                  carries its key id for candidate extraction, which has no length gate)
 *   bits 12:6  base quality (phred, clamped to 127)
 *   bits 20:13 mapping quality
 *   bits 31:21 key id: index of this entry's distinct indel key within its column (first-seen order)
 * key_meta (uint8) per distinct indel key: bits 1:0 kind (1 ins / 2 del), bit 2 = forward strand,
 *   bit 3 = longer than max_indel_length.
 * key_group (int32) per distinct key: index, within its column, of the strand-/anchor-case-merged allele the
 *   key belongs to for candidate extraction (insertions: anchor + sequence upper-cased; deletions: length),
 *   src/extract_candidates_calling.py:118-126.
 * col_ref bit 7 is set when the raw reference base is not A/C/G/T (such rows are skipped by candidate
 *   extraction, extract_candidates_calling.py:329-331, but not by tensor creation).
 * ---------------------------------------------------------------------------------------------- */
typedef struct cto_pack_view {
    int64_t         n_cols;
    int64_t         n_entries;
    int64_t         n_keys;
    const int32_t*  col_pos;   /* [n_cols]   1-based reference position, strictly increasing       */
    const uint8_t*  col_ref;   /* [n_cols]   bits 1:0 reference base code after evc_base_from (F1);
                                             bit 7: raw reference base is not A/C/G/T               */
    const int64_t*  col_off;   /* [n_cols+1] first entry of each column                             */
    const int32_t*  key_off;   /* [n_cols+1] first distinct indel key of each column                */
    const uint32_t* entries;   /* [n_entries]                                                       */
    const uint8_t*  key_meta;  /* [n_keys]                                                          */
    const int32_t*  key_group; /* [n_keys]                                                          */
} cto_pack_view;

typedef struct cto_pack cto_pack;   /* host-side pack incl. the key strings needed for alt_info   */

/* Parse `samtools mpileup --reverse-del --output-MQ --min-BQ 0` text (rows "chr\tpos\tN\tdepth\tbases\tBQ\tMQ")
 * into a pack.  ref_seq is the (raw, any case) reference covering [ref_start, ref_start+ref_len),
 * ref_start 1-based.  Replaces the text tokeniser of src/create_tensor_pileup_calling.py:120-144 and
 * the row handling at :472-497.  max_indel_length: shared/param.py max_indel_length (60). */
int cto_pack_from_mpileup(const char* text, size_t len, const char* ref_seq, int64_t ref_start,
                          size_t ref_len, int max_indel_length, cto_pack** out);
/* The pack producers (this one and cto_pack_from_bam*) cut a call over several threads of their own: up to 32, or CTO_PACK_THREADS from
 * the environment, or - for calls made from the CALLING thread from now on - n (0 = back to the default).  A caller that runs several
 * producers side by side gives each its share of the cores this way. */
void cto_set_pack_threads(int n);
/* BAM -> pack without the mpileup text: the columns `samtools mpileup --reverse-del --output-MQ
 * -r ctg:start-end --min-MQ <min_mq> --min-BQ 0 [-l bed] --excl-flags <excl_flags> [--max-depth N]` would print
 * (create_tensor_pileup_calling.py:426-446 runs that command), tokenised as cto_pack_from_mpileup would.  Needs the
 * BAM's .bai index (bai_path NULL = bam_path + ".bai").  start/end are 1-based inclusive; bed = n_bed sorted,
 * non-overlapping [begin, end) pairs in 0-based BED coordinates, or NULL.  PARITY UNPINNED against samtools (absent
 * from both boxes); rules and known deviations are listed at the top of csrc/bam.cpp. */
int cto_pack_from_bam(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                      const int64_t* bed, int64_t n_bed, const char* ref_seq, int64_t ref_start, size_t ref_len,
                      int excl_flags, int min_mq, int max_depth, int max_indel_length, cto_pack** out);
/* The same with the BGZF blocks inflated on the DEVICE (csrc/inflate.hip: one wavefront per block) instead of on the host cores,
 * which is what bounds cto_pack_from_bam.  Sequence for one chunk:
 *   cto_bam_chunk_span   -> the byte range [*file_begin, *file_end) of the BAM that holds every BGZF block with alignments
 *                           overlapping ctg:start-end (from the .bai; read it - plus CTO_BGZF_PAD bytes of padding - into memory)
 *   cto_bgzf_scan        -> the block table of those bytes: payload offset / size, inflated size, output slot (256-byte aligned);
 *                           returns the number of blocks, *out_bytes = size of the output buffer
 *   [copy bytes + table to the device]  cto_bgzf_inflate (asynchronous on `stream`; status[b] != 0 marks a malformed block)
 *   [copy the inflated bytes back]      cto_pack_from_bam_inflated: cto_pack_from_bam reading those blocks from memory (blocks that
 *                           are not in the table - the header at the start of the file - are still inflated on the host). */
#define CTO_BGZF_PAD 1024
#define CTO_BGZF_SLOT_PAD 64   /* bytes behind every block's inflated size in the output buffer (cto_bgzf_scan lays the slots out so) */
typedef struct cto_bgzf_block {
    uint64_t file_off;        /* offset of the block (its gzip header) in the BAM file                      */
    uint64_t in_off;          /* offset of its DEFLATE payload in the byte range handed to cto_bgzf_scan    */
    uint64_t out_off;         /* offset of its inflated bytes in the output buffer (multiple of 256; the next slot
                                 starts at least CTO_BGZF_SLOT_PAD bytes behind out_off + isize)              */
    uint32_t csize, isize;    /* payload bytes, inflated bytes                                              */
    uint32_t bsize, crc32;    /* whole block incl. header and trailer; CRC-32 of the inflated bytes (gzip trailer) */
} cto_bgzf_block;
int cto_bam_chunk_span(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                       int64_t* file_begin, int64_t* file_end);
int64_t cto_bgzf_scan(const uint8_t* bytes, size_t len, int64_t file_begin, cto_bgzf_block* blocks, int64_t cap, int64_t* out_bytes);
int cto_bgzf_inflate(const void* d_bytes, const cto_bgzf_block* d_blocks, int n_blocks, void* d_out, int* d_status, void* stream);
int cto_pack_from_bam_inflated(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                               const int64_t* bed, int64_t n_bed, const char* ref_seq, int64_t ref_start, size_t ref_len,
                               int excl_flags, int min_mq, int max_depth, int max_indel_length,
                               const uint8_t* inflated, size_t inflated_len, const cto_bgzf_block* blocks, int64_t n_blocks,
                               cto_pack** out);
/* Reads -> columns on the DEVICE (csrc/pileup.hip): the pack of cto_pack_from_bam built in HBM from the blocks cto_bgzf_inflate left
 * there - the alignment records do not come back to the host and no pack goes up (BASELINE.json north_star: the per-site read pileups
 * are staged on the GPU).  Sequence for one chunk: cto_bam_chunk_span, cto_bgzf_scan, copy up, cto_bgzf_inflate (as above), then
 *   cto_bam_record_starts -> the record boundaries the .bai names inside the span (virtual offsets, ascending): chunk starts and the
 *                            linear index's 16 kb windows; *tid = the contig's reference id
 *   cto_pileup_device     -> *dev_view: the pack's arrays in device memory owned by `ctx` (valid until its next call);
 *                            *host_lite: a host pack WITHOUT entries (col_pos, col_ref, key_off, key tables and the alt_info key
 *                            strings - what cto_alt_info* read); the caller frees it with cto_pack_free.
 *                            *fallback = 1 (CTO_OK, nothing built): the chunk holds what this path does not do - paired reads, reference
 *                            skips (N), at least max_depth accepted reads, a column deeper than 2048 or with more than 64 distinct indel
 *                            keys - use cto_pack_from_bam[_inflated].  h_blocks: the block table on the HOST.  Synchronises `stream`
 *                            (sizes come back twice).  Every block's CRC-32 is checked on the device first.  PARITY UNPINNED against samtools;
 *                            held bit-equal to cto_pack_from_bam (tests/test_gpu_pileup.py). */
/* `samtools view BAM ctg:start-end [-q min_mq]` without samtools (src/realign_reads.py:411-416 runs that command): the alignments
 * overlapping the region as SAM rows (no header) with the eleven mandatory fields and the HP:i tag when present.  Returns the number
 * of rows, *need = bytes of text; CTO_ENOMEM when cap is smaller (call again).  PARITY UNPINNED against samtools. */
int64_t cto_bam_view(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end, int min_mq,
                     char* buf, size_t cap, size_t* need);
typedef struct cto_dev_pileup cto_dev_pileup;
int  cto_dev_pileup_create(cto_dev_pileup** out);
void cto_dev_pileup_destroy(cto_dev_pileup* ctx);
int64_t cto_bam_record_starts(const char* bam_path, const char* bai_path, const char* ctg_name, int64_t start, int64_t end,
                              int64_t file_begin, int64_t file_end, uint64_t* voffs, int64_t cap, int32_t* tid);
int cto_pileup_device(cto_dev_pileup* ctx, const void* d_inflated, const cto_bgzf_block* h_blocks, int64_t n_blocks,
                      const uint64_t* rec_voffs, int64_t n_starts, int32_t tid, int64_t start, int64_t end, const int64_t* bed,
                      int64_t n_bed, const char* ref_seq, int64_t ref_start, size_t ref_len, int excl_flags, int min_mq,
                      int max_depth, int max_indel_length, void* stream, cto_pack_view* dev_view, cto_pack** host_lite, int* fallback);
/* mpileup TEXT -> pack on the DEVICE (csrc/tokenise.hip): what cto_pack_from_mpileup builds - the tokeniser of decode_pileup_bases
 * (src/create_tensor_pileup_calling.py:120-144) and the row handling of :465-532 - born in HBM from the text as `samtools mpileup --output-MQ`
 * prints it: one lane per row, the single forward pass of the host tokeniser.  `text` is HOST memory (copied through the context's page-locked
 * buffer unless it IS that buffer: cto_dev_tokeniser_buffer(ctx, len) hands out room to read a file into).  *dev_view: the pack's arrays in device
 * memory owned by `ctx` (valid until its next call); *host_lite: a host pack WITHOUT entries (col_pos, col_ref, key_off, key tables, alt_info key
 * strings), freed with cto_pack_free.  *fallback = 1 (CTO_OK, nothing built): the text holds what the single pass declines (another field
 * count, a short quality string, '\r', a non-printable byte, an indel or '^' running into the field's end, more than 32 indel-carrying read-bases
 * in one row, an empty row, no final '\n', rows out of position order, a position outside [ref_start, ref_start + ref_len)) - the caller runs
 * cto_pack_from_mpileup, which defines the behaviour and words the errors.  Synchronises `stream` (sizes come back three times).  Bit-equal to
 * cto_pack_from_mpileup, array for array and key string for key string (tests/test_gpu_tokenise.py). */
typedef struct cto_dev_tokeniser cto_dev_tokeniser;
int   cto_dev_tokeniser_create(cto_dev_tokeniser** out);
void  cto_dev_tokeniser_destroy(cto_dev_tokeniser* ctx);
char* cto_dev_tokeniser_buffer(cto_dev_tokeniser* ctx, size_t len);
int   cto_tokenise_device(cto_dev_tokeniser* ctx, const char* text, size_t len, const char* ref_seq, int64_t ref_start, size_t ref_len,
                          int max_indel_length, void* stream, cto_pack_view* dev_view, cto_pack** host_lite, int* fallback);
/* test / tool aid: n bytes of device memory to the host (synchronous hipMemcpy); no reference counterpart */
int cto_device_read(const void* d_src, void* h_dst, size_t n);
/* Build a pack from caller-made arrays (synthetic generators, BAM readers); key strings are the
 * alt_info keys ("IACG", "DACGT") concatenated, key_str_off[n_keys+1]. Arrays are copied. */
int cto_pack_from_arrays(const cto_pack_view* host_view, const int64_t* key_str_off,
                         const char* key_str, cto_pack** out);
int cto_pack_view_of(const cto_pack* p, cto_pack_view* host_view);
/* alt_info key string of distinct key `k` (global index); returns its length. */
int cto_pack_key_string(const cto_pack* p, int64_t k, const char** s);
void cto_pack_free(cto_pack* p);

/* ------------------------------------------------------------------------------------------------
 * Featurisation (src/create_tensor_pileup_calling.py).
 * ---------------------------------------------------------------------------------------------- */
#define CTO_COLVEC_STRIDE 72          /* int16 per column: [2 passes][36] (34 channels + 2 pad)   */

/* Stage A: one 34-channel vector per column for the AFF pass (read-bases with BQ >= min_bq, as
 * `samtools --min-BQ min_bq` would print) and the NEG pass (all read-bases), in one sweep.
 * Replaces decode_pileup_bases (create_tensor_pileup_calling.py:95-233), called twice per site by the
 * reference (run_clairs_to:1228-1271).
 *   colvec   dev [n_cols][2][36] int16   pass 0 = AFF, pass 1 = NEG
 *   coldepth dev [n_cols][2]     int32   `depth` of F4 (MQ>=20, N / over-long indels excluded)
 *   keycnt   dev [n_keys]        uint32  low 16 = AFF count, high 16 = NEG count of each distinct key (written by the
 *                                        call, no initialisation needed)
 * Limit: column depth <= 32767 (CTO_EUNSUPPORTED is raised by the pack builders). */
int cto_featurize_columns(const cto_pack_view* dev_pack, int min_bq, int16_t* colvec, int32_t* coldepth,
                          uint32_t* keycnt, void* stream);

/* Stage B: 33x34 window per candidate + coverage rescale, both passes.
 * Replaces the window assembly of create_tensor_pileup_calling.py:536-570 and the rescale of
 * clairs/predict.py:172-207 (x * (min_rescale_cov/depth) in double, cast to float, when depth > min_rescale_cov).
 *   site_pos  dev [n_sites] int32   1-based candidate positions
 *   x_aff/x_neg dev [n_sites][33][34] float   network inputs (rescaled); may be NULL
 *   raw_aff/raw_neg dev [n_sites][33][34] int16  un-rescaled tensors (the reference's text tensor); may be NULL
 *   site_info dev [n_sites][12] int32: {centre column index or -1, depth_aff, depth_neg, flags,
 *             fwd A,C,G,T, rev A,C,G,T} -- strand counts per predict.py:626-642 (true ref count restored)
 *             flags bit0: site skipped by the reference (no mpileup row at the candidate, or window
 *             start < 1: create_tensor_pileup_calling.py:542,552)
 *   sitefirst dev [n_sites][2][4] int32  per pass: first-seen entry index, within the candidate's own column, of bases
 *             A,C,G,T (either strand, MQ>=20) or 0x7fffffff -- the alt_info key order (F5); may be NULL
 *   keyfirst  dev [n_keys][2]     int32  per pass first-seen entry index of each distinct indel key OF THE CANDIDATE COLUMNS
 *             (0x7fffffff if none; entries of other columns are left untouched); may be NULL
 * min_bq is the AFF pass's gate (as in cto_featurize_columns); min_rescale_cov <= 0 disables the rescale. */
int cto_gather_windows(const cto_pack_view* dev_pack, const int16_t* colvec, const int32_t* coldepth,
                       const int32_t* site_pos, int64_t n_sites, int min_bq, int min_rescale_cov,
                       float* x_aff, float* x_neg, int16_t* raw_aff, int16_t* raw_neg,
                       int32_t* site_info, int32_t* sitefirst, int32_t* keyfirst, void* stream);
/* Both stages in ONE kernel, one workgroup per candidate: the window's entries are streamed once, the 33 column histograms live in LDS
 * and the tensors are written directly - the int16 column vectors of Stage A never go to HBM and back.  Same results as the two calls
 * above, for callers that need per-CANDIDATE outputs only:
 *   site_colvec dev [n_sites][CTO_COLVEC_STRIDE] int16  the candidate column's own vector (zeros without one); may be NULL
 *   keycnt      dev [n_keys] uint32                     written for the keys of the CANDIDATE columns only (others untouched)
 *   everything else as cto_gather_windows.  Windows of candidates closer than 33 bases recompute the columns they share. */
int cto_featurize_sites(const cto_pack_view* dev_pack, const int32_t* site_pos, int64_t n_sites, int min_bq, int min_rescale_cov,
                        float* x_aff, float* x_neg, int16_t* raw_aff, int16_t* raw_neg, int32_t* site_info, int16_t* site_colvec,
                        int32_t* sitefirst, uint32_t* keycnt, int32_t* keyfirst, void* stream);


/* Candidate extraction (src/extract_candidates_calling.py:55-169, 322-372) on the same pack: read-bases with
 * MQ >= min_mq and BQ >= min_bq (what `samtools mpileup --min-MQ --min-BQ` would print) are counted per column,
 *   depth = bases in ACGTacgt*#;  pass_depth = depth > min_coverage;
 *   pass_snv   = some non-reference base with count / depth >= snv_min_af and count >= alt_base_num;
 *   pass_indel = (select_indel) some merged indel allele with count / depth >= indel_min_af and count >= alt_base_num;
 * flags dev [n_cols] uint8: bit0 SNV candidate, bit1 indel candidate, bit2 pass_af (bits 3-5: see cto_extract_mark below); depth dev
 * [n_cols] int32. */
int cto_extract_candidates(const cto_pack_view* dev_pack, int min_mq, int min_bq, double snv_min_af,
                           double indel_min_af, double min_coverage, int alt_base_num, int select_indel,
                           uint8_t* flags, int32_t* depth, void* stream);
/* The candidate lists of extract_candidates_calling.py:433-446 from those flags, on the device and in position order: out_pos
 * dev [cap] int32 receives the 1-based positions of the columns whose flag has `bit` set (1 = SNV list, 2 = indel list) and whose
 * position lies in [lo, hi]; *n_out (dev int32) their number (positions beyond cap are counted, not written).  scratch: dev
 * int32 [ceil(n_cols / 256) + 1].  Stream-ordered; three small launches. */
int cto_candidate_positions(const cto_pack_view* dev_pack, const uint8_t* flags, int bit, int32_t lo, int32_t hi, int32_t* out_pos,
                            int64_t cap, int32_t* scratch, int32_t* n_out, void* stream);

/* The rest of extract_candidates_calling's modes, on the same flags (src/extract_candidates_calling.py:225-238, 249-260, 302, 347-383, 437-446,
 * 490-497).  cto_extract_candidates also sets, for a row that exists (some read passes min_mq and the reference base is A/C/G/T: bit5, 32),
 * bit3 (8) = some non-reference A/C/G/T read-base without an indel and bit4 (16, select_indel only) = some read-base carrying an indel.
 *   cto_extract_restrict  rows outside a BED: d_intervals = n_intervals sorted, merged, 0-based half-open [begin, end) pairs (dev int32);
 *                         a position p is inside when begin < p <= end - what `samtools mpileup -l` prints (:302) and what
 *                         is_region_in(tree, ctg, p - 1, p) accepts (:437-446).  Columns outside lose the bits of `clear`: 0xff = the row
 *                         does not exist (confident BED; depth, if not NULL, becomes 0), 2|16 = no indel candidate there.
 *   cto_extract_mark      sets the bits of `set` on the columns whose position is in d_pos (dev int32 [n_pos], sorted): bit6 (64) = a position of
 *                         --hybrid_mode_vcf_fn / --genotyping_mode_vcf_fn.  cto_candidate_positions then also lists a marked column that
 *                         fails the AF gates (no bit2) when it shows an alternative base (SNV list) / an indel (indel list), :374-383.
 *   cto_hybrid_info       per position of d_pos (sorted) a record rec[16] (dev int32): column index (-1: no row), depth, count A C G T I D,
 *                         first-seen read index A C G T I D, first key of the column, the column's flags; with select_indel the merged
 *                         indel alleles are counted per group instead: gcnt / gfirst (dev [n_keys]) at [first key + group].
 *   cto_hybrid_info_rows  host: the rows of `<ctg>.<chunk>_hybrid_info` (:352-354, 490-497) from those records (copied to the host) and the
 *                         pack's key strings; returns the text length (when > cap nothing was written). */
int cto_extract_restrict(const cto_pack_view* dev_pack, uint8_t* flags, int32_t* depth, const int32_t* d_intervals, int n_intervals, int clear,
                         void* stream);
int cto_extract_mark(const cto_pack_view* dev_pack, uint8_t* flags, const int32_t* d_pos, int n_pos, int set, void* stream);
int cto_hybrid_info(const cto_pack_view* dev_pack, const uint8_t* flags, const int32_t* d_pos, int n_pos, int min_mq, int min_bq, int select_indel,
                    int32_t* rec, uint32_t* gcnt, int32_t* gfirst, void* stream);
int64_t cto_hybrid_info_rows(const cto_pack* p, const char* ctg, int64_t n_pos, const int32_t* pos, const int32_t* rec, int select_indel,
                             const uint32_t* gcnt, const int32_t* gfirst, char* buf, size_t cap);

/* Host: the reference's alt_info string "<depth>-<key count ...>-" for the column `col` of pass `pass`
 * (0 = AFF, 1 = NEG; create_tensor_pileup_calling.py:158-209), from stage-A outputs copied to the host.
 * colvec_col points at that column's [2][36] int16, colfirst_col at the site's [2][4] int32 row of `sitefirst`; keycnt / keyfirst are the whole
 * arrays.  Returns the string length (<= cap-1) or an error. */
int cto_alt_info(const cto_pack* p, int64_t col, int pass, const int16_t* colvec_col, int32_t depth,
                 const int32_t* colfirst_col, const uint32_t* keycnt, const int32_t* keyfirst,
                 char* buf, size_t cap);
/* The same for all sites of a chunk in one call: site_info / sitefirst as written by cto_gather_windows (host copies), colvec
 * the whole host copy [n_cols][2][36].  Strings are packed back to back into buf; offsets[n_sites + 1] delimits them (a site
 * without a centre column gets an empty string).  Returns the bytes used or a negative error code. */
int64_t cto_alt_info_batch(const cto_pack* p, int64_t n_sites, const int32_t* site_info, int pass, const int16_t* colvec,
                           const int32_t* sitefirst, const uint32_t* keycnt, const int32_t* keyfirst, char* buf, size_t cap,
                           int64_t* offsets);
/* The same with the candidate columns' vectors gathered on the device: site_colvec[n_sites][CTO_COLVEC_STRIDE], row i = the column
 * vector of site i's candidate column (a per-chunk device-to-host copy of 144 B per site instead of 144 B per pack column). */
int64_t cto_alt_info_batch_sites(const cto_pack* p, int64_t n_sites, const int32_t* site_info, int pass, const int16_t* site_colvec,
                                 const int32_t* sitefirst, const uint32_t* keycnt, const int32_t* keyfirst, char* buf, size_t cap,
                                 int64_t* offsets);


/* ------------------------------------------------------------------------------------------------
 * Models (clairs/model.py).  Weights are handed over by state_dict name; data is host fp32.
 * ---------------------------------------------------------------------------------------------- */
typedef struct cto_weights cto_weights;
typedef struct cto_model   cto_model;

cto_weights* cto_weights_new(void);
int  cto_weights_add(cto_weights* w, const char* name, const float* data, int64_t numel);
void cto_weights_free(cto_weights* w);

typedef struct cto_cvt_cfg {
    int emb_dim[3];   /* s{1,2,3}_emb_dim, e.g. 16, 64, 128  (clairs/predict.py:520-553)        */
    int heads[3];     /* s{1,2,3}_heads,   e.g. 1, 3, 4                                         */
    int depth[3];     /* s{1,2,3}_depth,   e.g. 1, 2, 3                                         */
    int n_out;        /* 4 (CvT: a c g t) or 6 (CvT_Indel: a c g t i d)                         */
} cto_cvt_cfg;

/* clairs.model.CvT / CvT_Indel (clairs/model.py:150-384): eval-mode forward. */
int cto_cvt_create(const cto_weights* w, const cto_cvt_cfg* cfg, cto_model** out);
/* clairs.model.BiGRU_NACGT / BiGRU_NACGT_Indel (clairs/model.py:387-560); n_out = 4 or 6. */
int cto_bigru_create(const cto_weights* w, int n_out, cto_model** out);
/* The same with the arithmetic of the GEMM operands chosen by the caller instead of by the process environment
 * (the reference has one arithmetic, fp32: clairs/predict.py:512-568 builds the modules and never casts them).
 * CTO_SPLIT_ENV = what cto_*_create do: fp32 unless CTO_CVT_SPLIT / CTO_GRU_SPLIT name "f16" or "bf16";
 * CTO_SPLIT_F32 = fp32 MFMA whatever the environment says; CTO_SPLIT_F16 / _BF16 = every GEMM operand written as
 * hi + lo 16-bit halves, three MFMA passes per product, fp32 accumulation (the opt-in side channel of DESIGN.md 6). */
#define CTO_SPLIT_ENV  (-1)
#define CTO_SPLIT_F32  0
#define CTO_SPLIT_F16  1
#define CTO_SPLIT_BF16 2
int cto_cvt_create_ex(const cto_weights* w, const cto_cvt_cfg* cfg, int split_mode, cto_model** out);
int cto_bigru_create_ex(const cto_weights* w, int n_out, int split_mode, cto_model** out);
/* One-blob hand-over: `packed` host fp32 = every tensor of the module's state_dict() concatenated in state_dict order
 * (clairs/model.py:150-560 as `torch.save`d at clairs/predict.py:513-517; the integer `num_batches_tracked` entries are left
 * out) - `torch.cat([v.flatten() for v in model.state_dict().values() if v.is_floating_point()])`.  This is the
 * `packed_weights` argument of the torch custom ops clairsto::cvt_forward / clairsto::bigru_forward.
 * CTO_EMISSING when numel does not match the manifest. */
int cto_cvt_create_packed(const float* packed, int64_t numel, const cto_cvt_cfg* cfg, cto_model** out);
int cto_bigru_create_packed(const float* packed, int64_t numel, int n_out, cto_model** out);
/* The manifest the packed creators walk: one "name<TAB>numel<NL>" line per tensor, kind 0 = CvT (cfg, its n_out) or
 * 1 = BiGRU (n_out; cfg ignored).  Returns the bytes needed including the terminating NUL (the text is written only when
 * cap is large enough), or a negative error code. */
int64_t cto_model_manifest(int kind, const cto_cvt_cfg* cfg, int n_out, char* buf, size_t cap);
/* x dev [B][33][34] float -> logits dev [n_out][B][2] float (post-SELU, pre-softmax), exactly the tuple
 * `model(x)` returns at clairs/predict.py:646-658. */
int cto_model_forward(cto_model* m, const float* x, int64_t B, float* logits, void* stream);
/* The same from the un-rescaled int16 tensor (raw_aff / raw_neg of cto_featurize_sites) - the form the reference keeps its tensors in
 * (create_tensor_pileup_calling.py writes integer text; clairs/predict.py:172-207 rescales on load): x_raw dev [B][33][34] int16,
 * site_info dev [B][12] as cto_featurize_sites writes it (the depth of pass `which` = 0 AFF / 1 NEG at [1 + which]), min_rescale_cov as
 * given to the tensor kernel (<= 0: no rescale).  The first layer's loader converts - float(double(v) * min_rescale_cov / depth), the
 * tensor kernel's own expression, so the logits equal cto_model_forward's on the fp32 tensor bit for bit - and the fp32 tensor is
 * never written or read.  Handles whose first layer has no int16 loader (split operands, CTO_GRU_ROT=0) expand it once inside the call. */
int cto_model_forward_raw(cto_model* m, const int16_t* x_raw, const int32_t* site_info, int which, int min_rescale_cov, int64_t B,
                          float* logits, void* stream);
/* algorithmic multiply-accumulate count per site of this model (for roofline accounting). */
int64_t cto_model_macs_per_site(const cto_model* m);
int  cto_model_n_out(const cto_model* m);
void cto_model_destroy(cto_model* m);
/* Live kernel timing for roofline accounting: when enabled (1), the dominant kernel of the model (BiGRU: the
 * layer-2 recurrent kernel; CvT: the whole forward) is bracketed by HIP events on the launch stream; 2 brackets BiGRU layer 1 as
 * well (an event pair costs the stream ~6 us of dispatch gap per launch, so the timed region of bench.py uses 1).
 * cto_model_profile_read waits for the recorded events, returns the number of launches measured since the
 * last read, writes their mean duration in milliseconds and the algorithmic multiply-accumulates of ONE site
 * in that kernel (launch MACs = per-site MACs x batch). */
int cto_model_profile(cto_model* m, int enable);
int cto_model_profile_read(cto_model* m, double* mean_ms, int64_t* macs_per_site);
/* The same for one more kernel of the model: stage 0 = the above, stage 1 = BiGRU layer 1 (bracketed in mode 2).
 * Its events are kept until read; stage 1 of a CvT handle is CTO_EINVAL. */
int cto_model_profile_read_stage(cto_model* m, int stage, double* mean_ms, int64_t* macs_per_site);

/* ------------------------------------------------------------------------------------------------
 * Posterior / decision / quality (clairs/call_variants.py:154-304, 79-88), fused with the 2-way
 * softmax of clairs/predict.py:659-684 and the "{:0.8f}" round trip of the probability text seam
 * (predict.py:114-152 -> call_variants.py:803-829).
 *   aff_logits, neg_logits dev [K][B][2] float
 *   lik   dev [K][10][10] double   likelihood matrices (call_variants.py:661-664 / 717-722)
 *   edges dev [2K][11]   double    bin edges with 0 prepended and 1 appended, order a,na,c,nc,...
 *   probs dev [B][2K][2] float     softmax outputs, order a c g t [i d] na nc ng nt [ni nd]; may be NULL
 *   post  dev [B][K] double        posterior per base
 *   decision dev [B][4] int32      {argmax (np.argmax: first maximum, a NaN first), flags, 0, 0}; flags bit 0 = a bin index
 *                                  was clamped (the reference raises IndexError there: a probability printed as 1.00000000
 *                                  or 0.00000000), bit 1 = the winning posterior is NaN (0/0; only together with bit 0):
 *                                  no row can be formatted for that site; bit 2 = QUAL sits within 1e-10 of a 4-decimal
 *                                  rounding boundary, [2..3] then hold the bits of the winning posterior (cto_qual_finalize)
 *   qual  dev [B] double           quality_score_from(max posterior), rounded to 4 dp (call_variants.py:79-88)
 * ---------------------------------------------------------------------------------------------- */
int cto_posterior(const float* aff_logits, const float* neg_logits, int K, int64_t B,
                  const double* lik, const double* edges, float* probs, double* post,
                  int32_t* decision, double* qual, void* stream);

/* Host half of QUAL (clairs/call_variants.py:79-88: math.log is the host C library's log).  The device's log() may differ from
 * the host's in the last bit, which changes round(q, 4) only for a q within ~1e-14 of a ...5 boundary; cto_posterior flags
 * every site within 1e-10 of one (about two in a million).  This call, on HOST copies of decision / qual, re-evaluates the
 * flagged sites with the host libm, clears the flag and the posterior bits - afterwards decision[i] = {argmax, flags (bits 0-1),
 * 0, 0} and qual[i] is bit for bit what the reference prints on this machine.  Idempotent; returns the number of sites rewritten.
 * cto_vcf_rows_batch applies the same rule itself. */
int64_t cto_qual_finalize(int32_t* decision, double* qual, int64_t n);
/* How many of the B sites of a DEVICE decision array wait for that host half (flag bit 2): *count (device int32) is zeroed and
 * filled on `stream`.  A caller that wants final values without downloading 16 B per site every time (the torch operator) reads
 * these four bytes and runs cto_qual_finalize on host copies only when they are non-zero - about one 4096-site call in a hundred. */
int cto_qual_pending(const int32_t* decision, int64_t B, int32_t* count, void* stream);

/* Only the 2-way softmax of clairs/predict.py:659-684: probs dev [B][2K][2] in the order the probability text
 * rows use (a c g t [i d] na nc ng nt [ni nd]). */
int cto_softmax_probs(const float* aff_logits, const float* neg_logits, int K, int64_t B, float* probs, void* stream);
/* nn.Softmax(dim=1) over n rows of two logits (dev [n][2] -> dev [n][2]): what a module built with apply_softmax=True applies to each
 * head before returning (clairs/model.py:255-259, :461-465).  Same arithmetic as cto_softmax_probs. */
int cto_softmax_pairs(const float* logits, int64_t n, float* out, void* stream);

/* Same epilogue entered at the probability text seam (clairs/call_variants.py:798-829 parses the rows that
 * clairs/predict.py:114-152 wrote): p1 dev [B][2K] double = the second number of each "p0 p1" field, in the
 * order a c g t [i d] na nc ng nt [ni nd], already rounded to 8 decimals by the producer. */
int cto_posterior_from_probs(const double* p1, int K, int64_t B, const double* lik, const double* edges,
                             double* post, int32_t* decision, double* qual, void* stream);

/* ------------------------------------------------------------------------------------------------
 * VCF data rows of a whole chunk in one call: the string work left after the device epilogue -
 * clairs/call_variants.py:135-150 (alt_info), 306-365 (allele ranking, ALT / REF), 367-380 (drop rules),
 * 401-415 (AF, GT), 67-76 (FILTER), 588-618 (INFO / FORMAT) and VcfWriter.write_row (shared/vcf.py:144-185).
 *   pos[n] 1-based; centre[n] reference base of each row (clairs/predict.py:415);
 *   alt_buf + alt_off[n+1]: AFF alt_info strings as cto_alt_info_batch packs them; site_info[n][12] from cto_gather_windows
 *   ([3] bit 0 = skip: site without a tensor, or dropped by the caller because its raw centre is not in "ACGT",
 *   clairs/predict.py:219-228; [4..12) strand counts); decision[n][4] / qual[n] from cto_posterior;
 *   K = 4 (SNV mode) or 6 (indel mode); show_ref = --show_ref; qual_pass = --qual (< 0: no threshold).
 *   counts[4] out: rows written, sites processed, "low tumor coverage" events (call_variants.py:327-329), sites flagged clamped.
 * Rows are '\n'-terminated in buf.  Returns the bytes used, CTO_ENOMEM when cap is too small, CTO_EINVAL on malformed input. */
int64_t cto_vcf_rows_batch(const char* chrom, int64_t n, const int32_t* pos, const char* centre, const char* alt_buf,
                           const int64_t* alt_off, const int32_t* site_info, const int32_t* decision, const double* qual,
                           int K, int show_ref, double qual_pass, char* buf, size_t cap, int64_t* counts);

/* Candidate BED chunk file (text of `<ctg>.<i>_<n>_snv`, extract_candidates_calling.py:450-488) -> window centres, as
 * create_tensor_pileup_calling.py:347-370 derives them: out[0..min(n, cap)) in file order (sort + de-duplicate to get the
 * reference's dict keys), span[0] / span[1] = ctg_start / ctg_end of :359-360, *has_types = 1 when a row carries the
 * optional fourth column.  Returns the number of rows of contig `ctg`, or a negative error code. */
int64_t cto_bed_centres(const char* text, size_t len, const char* ctg, int32_t* out, int64_t cap, int64_t* span, int* has_types);

/* ------------------------------------------------------------------------------------------------
 * All candidate chunks of a run, files in, files out: what run_clairs_to:1228-1308 / 1562-1647 does per chunk with four
 * commands (create_tensor_pileup_calling x 2, predict --pileup, call_variants) as ONE call for the whole chunk list -
 * producer threads (BED -> centres, reference slice, column pack, upload on their own streams), the calling thread launching
 * featurisation + both networks + epilogue on `stream`, writer threads (alt_info strings, VCF records, p_<chunk>.vcf).
 * BED and pileup-text paths ending in .gz are inflated (as the reference's readers gzip.open them); the reference FASTA must be plain.
 * A chunk without records leaves no file (call_variants.py:859-867).
 * ---------------------------------------------------------------------------------------------- */
typedef struct cto_chunk_job {
    const char* ctg_name;       /* --ctg_name                                                                  */
    const char* bed_path;       /* --candidates_bed_regions: the chunk's candidate BED                          */
    const char* mpileup_path;   /* `samtools mpileup --min-BQ 0 ...` text of the chunk, or NULL: read bam_path  */
    const char* bam_path;       /* --tumor_bam_fn (+ .bai), through cto_pack_from_bam                           */
    const char* vcf_path;       /* --call_fn                                                                    */
    /* REGION job (bed_path == NULL): candidate extraction is an internal product of the run - STEP 1 of the reference
     * (src/extract_candidates_calling.py:172-503, run_clairs_to:1196-1220) fused in front of tensor creation.  The region
     * [region_start, region_end] (1-based, the --ctg_start / --ctg_end of extract_candidates_calling or its chunk_id split,
     * :252-281) is piled up ONCE (all positions of [start - 33 - 17, end + 33 + 17]: the reference's own read region :289-292 plus
     * the flanks of the windows at its edges) from bam_path (or from mpileup_path: `samtools mpileup --reverse-del --output-MQ
     * --min-MQ 0 --min-BQ 0 -r` text of that range); the gates of cto_run_cfg run on that pack in HBM (cto_extract_candidates; every
     * row of [start - 33, end + 33] takes part, as in the reference), the SNV list (K = 4) or the indel list (K = 6) becomes
     * the chunk's candidate sites, and the same pack feeds tensor creation, the networks and the epilogue. */
    int64_t region_start, region_end;
    const char* candidates_path; /* REGION job, optional: the candidates as the rows of the reference's `<ctg>.<chunk>_<i>_<n>_snv|_indel`
                                    BED chunk files (`ctg \t max(x-17,1) \t x+17`, :450-488), one file for the whole region          */
    /* REGION job, the other modes of extract_candidates_calling - each an interval test or a marker on the flags in HBM (cto_extract_restrict /
     * cto_extract_mark / cto_hybrid_info), between the gates and the compaction of the candidate list: */
    const int32_t* confident_intervals; /* --bed_fn (:249-260, 302: `samtools mpileup -l`): host int32 [2 n], sorted, merged, 0-based half-open
                                    [begin, end) rows of the job's contig; a position p exists when begin < p <= end.  Read when
                                    restrict_to_confident != 0 (n = 0 then means: no row of the region exists, no candidate)          */
    int64_t n_confident_intervals;
    int restrict_to_confident;
    const int32_t* known_pos;     /* --hybrid_mode_vcf_fn / --genotyping_mode_vcf_fn (:225-238, 347-349, 370-383): host int32 [n], sorted
                                    positions of the VCF's records of this contig; one that has a row and shows an alternative base
                                    (K = 4) / an indel (K = 6) is a candidate whether or not it passes the AF gates                    */
    int64_t n_known_pos;
    const char* hybrid_info_path; /* optional: the rows of `<ctg>.<chunk>_hybrid_info` (:352-354, 490-497) of the known positions inside
                                    [region_start - 33, region_end + 33], written here (an empty file when there are none)           */
} cto_chunk_job;
typedef struct cto_run_cfg {
    cto_model*    aff;          /* CvT / CvT_Indel                                                              */
    cto_model*    neg;          /* BiGRU_NACGT / BiGRU_NACGT_Indel                                              */
    const double* d_lik;        /* dev, as cto_posterior takes them                                             */
    const double* d_edges;
    int    K;                   /* 4 or 6                                                                       */
    int    min_bq;              /* AFF pass gate (--min_bq / the platform's default)                            */
    int    min_rescale_cov;     /* --min_rescale_cov (50); <= 0: no rescale                                     */
    int    max_indel_length;    /* shared/param.py max_indel_length (60)                                        */
    int    max_depth;           /* BAM input: --max-depth of the pileup (8000)                                  */
    int    neg_reads_aff;       /* ilmn: the NEG network reads the AFF tensors (run_clairs_to:1248-1252)        */
    int    show_ref;            /* --show_ref                                                                   */
    int    verbose;             /* the reference's per-chunk console lines (stdout / stderr)                    */
    double qual_pass;           /* --qual (< 0: no threshold)                                                   */
    const char* ref_fa;         /* --ref_fn, uncompressed, with its .fai                                        */
    const char* vcf_header;     /* everything before the first record, incl. the #CHROM line                    */
    int    producers, writers;  /* threads (>= 1)                                                               */
    int    depth;               /* chunks in flight (0: producers + writers + 2); each holds a pack on both sides */
    int    inflate_cus;         /* BAM input: > 0 = up to inflate_jobs chunks at a time have their BGZF blocks inflated on the device
                                   (cto_bgzf_inflate) on streams confined to the first inflate_cus compute units, the others on the
                                   host cores as cto_pack_from_bam does; 0 = host only.  Same packs either way.               */
    int    inflate_jobs;        /* `stream` of cto_run_chunks should be a non-blocking stream: the legacy default stream synchronises
                                   with the CU-masked (blocking) inflate streams and the two exclude each other; NULL = the call
                                   creates one of its own, ordered behind the default stream's work so far.                    */
    int    pack_threads;        /* threads one producer's tokeniser / BAM decoder call may use (0: CTO_PACK_THREADS or the library's
                                   default of up to 32): producers x pack_threads should stay near the usable cores            */
    const char*   samtools;     /* BAM input: NULL = the built-in reader; else this program (PATH is searched) is run per chunk as
                                   `samtools mpileup --reverse-del --output-MQ -r ctg:s-e --min-MQ 0 --min-BQ 0 -l <bed> --excl-flags 2316
                                   [--max-depth samtools_max_depth] <bam>` (create_tensor_pileup_calling.py:426-446, with --min-BQ 0)
                                   and its text tokenised                                                                      */
    int    samtools_max_depth;  /* > 0: passed as --max-depth                                                                 */
    cto_model*    aff2;         /* optional second pair of handles of the same weights (NULL: none): with them consecutive chunks  */
    cto_model*    neg2;         /* alternate between two compute streams, and the next chunk's first round of workgroups fills the
                                   CUs this chunk's last round leaves idle (chunk sizes that are not a multiple of 4096 sites:
                                   the recurrent kernels put 32 sites on a CU)                                                  */
    int    device_pileup;       /* BAM input with inflate_cus > 0: 1 = the chunks that go through the device inflate are piled up there
                                   too (cto_pileup_device: the records stay in HBM, the pack is built in HBM); a chunk that path does
                                   not take (paired reads, ...) is piled up on the host from the device-inflated blocks.  Same packs.  */
    /* gates of REGION jobs (extract_candidates_calling's options; ignored by BED jobs): */
    int    extract_min_mq;      /* --min_mq (20)                                                                               */
    int    extract_min_bq;      /* --min_bq of STEP 1 (the platform's min_bq, run_clairs_to:1201)                               */
    int    alt_base_num;        /* --alternative_base_num (3)                                                                  */
    double snv_min_af;          /* --snv_min_af (0.05)                                                                         */
    double indel_min_af;        /* --indel_min_af (ONT 0.1, others 0.05); used when K = 6                                      */
    double min_coverage;        /* --min_coverage (4): depth must EXCEED it                                                    */
    const char* indel_regions_bed; /* --call_indels_only_in_these_regions (NULL: none): K = 6 REGION jobs keep an indel candidate only
                                   when [pos - 1, pos) overlaps a row of its contig (0-based half-open rows, start == end widened by
                                   one; a BED without rows of the contig filters nothing) - extract_candidates_calling.py:437-446   */
    int    device_tokenise;     /* mpileup-text input (mpileup_path, or the samtools child's output): 1 = the text goes up as it is and the
                                   pack is built in HBM (cto_tokenise_device); a text that path declines is tokenised on the host
                                   (cto_pack_from_mpileup).  Same packs.                                                         */
    int    indel_bed_superseded; /* the user gave --bed_fn (run_clairs_to's --bed_fn_source): indel_regions_bed filters nothing then
                                   (extract_candidates_calling.py:438)                                                            */
} cto_run_cfg;
typedef struct cto_run_stats {
    int64_t candidates;                                /* candidate positions read from the BED chunks / extracted from the regions */
    int64_t sites, rows, low_coverage, clamped;        /* sums of cto_vcf_rows_batch's counts                    */
    double  seconds;                                   /* wall clock of the call                                */
    double  produce_s, finish_s;                       /* thread-seconds summed over the producer / writer threads */
    double  launch_s, launcher_wait_s;                 /* the calling thread: launching, waiting for a producer */
    double  pack_s, upload_s;                          /* parts of produce_s: tokenising / BAM decoding, host-to-device copies */
    double  device_s;                                  /* HIP-event time from a chunk's first kernel to its last copy, summed    */
    int64_t device_piled;                              /* chunks whose pack was built on the device (cto_pileup_device) */
    int64_t device_inflated;                           /* BAM chunks whose blocks were inflated on the device                    */
    int64_t device_tokenised;                          /* text chunks whose pack was built on the device (cto_tokenise_device)   */
} cto_run_stats;
int cto_run_chunks(const cto_run_cfg* cfg, const cto_chunk_job* jobs, int64_t n_jobs, void* stream, cto_run_stats* stats);
/* cto_run_chunks keeps its per-chunk buffers (device, page-locked host, events) for the next call on the same device; this
 * frees them.  Not needed before process exit. */
int cto_run_release(void);

/* ------------------------------------------------------------------------------------------------
 * Long-read post-calling filters (src/haplotype_filtering.py:344-707): the read-level evidence of
 * every call of ONE mpileup job, from the nine-column text of
 *   samtools mpileup --min-MQ q --min-BQ q --excl-flags 2316 [-l bed] -r ctg:lo-hi --output-MQ --output-QNAME --output-extra HP
 * (haplotype_filtering.py:336-345).  Host code (no device work: thousands of calls, <= 201 columns each).
 *   ref_seq                  upper-cased reference of [region_lo, region_lo + ref_len) (`samtools faidx ctg:lo-hi`, :1096-1101)
 *   pos[n]                   the calls, 1-based;  fields + field_off[n+1]: per call "REF\tALT\tHETERO_INFO\tHOMO_INFO" back to
 *                            back (HETERO / HOMO_INFO = "pos-ALT,pos-ALT,..." of the phased germline variants within
 *                            +-flanking, :1012-1021);  af[n] (1.0 when unknown)
 *   flanking                 --flanking (100);  max_co_exist_read_num = --min_alt_coverage (2);  disable_rse =
 *                            --disable_read_start_end_filtering
 *   flags[n][9] uint8        1 = True of: phaseable, pass_hetero, pass_homo, pass_read_start_end, pass_bq, pass_mq, pass_co_exist,
 *                            pass_hetero_both_side, pass_sequence_entropy (the columns of the reference's per-call output line, :588-592)
 *   strand[n][4] int64       a0, r0, a1, r1: the 2x2 table of Fisher's exact test (:575-582); the caller evaluates the p-value
 *                            in exact integer arithmetic as the reference does (:60-97) and derives pass_strand_bias / pass_hap
 * ---------------------------------------------------------------------------------------------- */
int cto_haplotype_filter(const char* text, size_t len, const char* ref_seq, int64_t region_lo, size_t ref_len, int64_t n,
                         const int32_t* pos, const char* fields, const int64_t* field_off, const double* af, int flanking,
                         int max_co_exist_read_num, int disable_rse, uint8_t* flags, int64_t* strand);

/* ------------------------------------------------------------------------------------------------
 * Illumina realignment filter (SURVEY.md 8f #4b; src/realign_variants.py, src/realign_reads.py, the C++ under src/realign/).
 * Host code: ~1 k low-QUAL calls per run, a window of <= 1000 reads each.
 *
 * cto_realign_reads replaces the reference's native entry point
 *   struct_str_arr* realign_reads(char* seqs[], int* positions, char* cigars[], char* reference, char* haplotypes,
 *                                 int ref_start, int ref_prefix, int ref_suffix, int read_size)   (src/realign/realigner.cpp:782-857,
 *   bound with ctypes at src/realign_reads.py:582-591): the reads of one window are re-aligned through the candidate haplotypes
 *   (blank-separated, each = reference prefix + consensus + reference suffix) and come back as 0-based positions and CIGAR text over
 *   S X I D (X = aligned, as the reference prints it; the caller maps X to M).  A read nothing aligned keeps positions[i] / cigars[i].
 *   out_positions[n]; CIGAR strings '\0'-terminated back to back in cigar_buf, cigar_off[n+1] their offsets.  CTO_ENOMEM when
 *   cigar_cap is too small, CTO_EINVAL for a haplotype shorter than 32 bases (the reference indexes past the string there).
 * cto_ssw_align: the Smith-Waterman the above is built on, alone - SSW's C++ `Aligner::Align(query, filter, &alignment)` with the
 *   reference sequence `ref` (src/realign/ssw_cpp.cpp:302-337): score 4 / -6, gap 8 / 2; *score = 0 and an empty CIGAR when nothing aligns.
 * cto_dbg_consensus replaces  get_consensus(char* reference, char* reads ","-joined, char* low-BQ positions " "/","-joined, int n)
 *   (src/realign/debruijn_graph.cpp:432-470, bound at src/realign_reads.py:532-536): candidate haplotypes of a window, sorted,
 *   '\0'-separated in buf; returns their number.  lowbq / lowbq_off[n+1]: per read the 0-based positions with BQ < 15 (may be NULL).
 *   *used = bytes needed.  PARITY UNPINNED (the reference needs Boost.Graph, absent here): see csrc/debruijn.cpp.
 * The reference's own symbol names and struct layouts are exported by two one-file shims built next to the library
 * (clairs_to_amd/realign/realigner.so, debruijn_graph.so; csrc/ref_abi_*.cpp) for `ctypes.cdll.LoadLibrary` at
 * src/realign_reads.py:70-71.
 * ---------------------------------------------------------------------------------------------- */
int cto_realign_reads(int n_reads, const char* const* seqs, const int32_t* positions, const char* const* cigars,
                      const char* reference, const char* haplotypes, int32_t ref_start, int32_t ref_prefix, int32_t ref_suffix,
                      int32_t* out_positions, char* cigar_buf, size_t cigar_cap, int64_t* cigar_off);
int cto_ssw_align(const char* ref, const char* query, int32_t* score, int32_t* ref_begin, char* cigar_buf, size_t cigar_cap);
/* Every window of a run in one call (csrc/realign_batch.hip).  The reference calls realign_reads(...) (src/realign/realigner.cpp:782-857,
 * bound at src/realign_reads.py:582-591) once per window from one Python process per low-QUAL call (src/realign_variants.py:73-110);
 * a job is the argument list of one such call plus the caller's output buffers (as cto_realign_reads: out_positions[n_reads],
 * cigar_off[n_reads + 1], the CIGARs NUL-terminated in cigar_buf).  where = CTO_REALIGN_HOST: the windows are dealt to host_threads
 * workers (<= 0: cto_set_realign_threads' value), each running what cto_realign_reads runs.  where = CTO_REALIGN_DEVICE: the k-mer
 * fast pass (realigner.cpp:129-229) of every (window, haplotype) and both striped Smith-Waterman passes (ssw.c:118-529 as
 * ssw_align drives them, :781-830) of every haplotype / reference and unplaced-read / haplotype pair run as kernels on `stream` (and streams forked from it)
 * (HIP device current to the caller), the banded tracebacks (ssw.c:531-741) the windows will need - every haplotype against the reference,
 * per unplaced read the pair it picks - as a third stage (k_banded); CIGAR composition stays on the host (host_threads workers).
 * Windows the device form does not take (a haplotype or the reference longer than 2 048 bases, a read longer than 512) run on the
 * host inside the same call.  Outputs are the reference's byte for byte either way (tests/test_gpu_realign.py holds both to
 * oracle/_ref).  jobs[i].status is that window's code; the return value is the first failing window's code. */
typedef struct cto_realign_job {
    int32_t n_reads;
    const char* const* seqs;           /* [n_reads] NUL-terminated read bases                                   */
    const int32_t* positions;          /* [n_reads] current 0-based alignment starts                            */
    const char* const* cigars;         /* [n_reads] current CIGARs                                              */
    const char* reference;             /* prefix + window + suffix                                              */
    const char* haplotypes;            /* candidate haplotypes, white-space separated (get_consensus' output)  */
    int32_t ref_start, ref_prefix, ref_suffix;
    int32_t* out_positions;            /* [n_reads]                                                             */
    char* cigar_buf; size_t cigar_cap; /* CIGAR text, NUL-terminated one after the other                        */
    int64_t* cigar_off;                /* [n_reads + 1] offsets into cigar_buf                                  */
    int32_t status;                    /* out                                                                   */
    const char* seqs_joined;           /* read when seqs == NULL: the n_reads strings back to back, each NUL-terminated - a host  */
    const char* cigars_joined;         /* language that holds them as one buffer need not build pointer arrays; same for cigars   */
} cto_realign_job;
typedef struct cto_realign_stats {
    int64_t windows, host_windows, reads, haplotypes;
    int64_t fast_pairs;                /* (haplotype, read) pairs of the device fast pass                       */
    int64_t sw_pairs, sw_cells;        /* device Smith-Waterman alignments and their ref x query cells          */
    double fast_pass_ms, sw_ms;        /* HIP-event time of the two stages' launches                            */
    double device_stage_ms, host_ms;   /* wall time up to / after the device stages (packing and copies included) */
    int64_t tracebacks, tracebacks_declined;   /* banded tracebacks sent to the device; those it left to the host     */
    double traceback_ms;               /* HIP-event time of their launches                                      */
} cto_realign_stats;
#define CTO_REALIGN_HOST   0
#define CTO_REALIGN_DEVICE 1
int cto_realign_windows(int n_jobs, cto_realign_job* jobs, int where, int host_threads, void* stream, cto_realign_stats* stats);
/* The Smith-Waterman stage of cto_realign_windows on its own: the two striped passes ssw_align runs per alignment (ssw.c:781-830: 8-bit
 * forward pass, 16-bit passes instead when it overflows at 249, backward pass over the reversed prefixes) for n independent alignments.
 * codes = base codes 0..4 (A C G T other; n_codes bytes), desc[k] = {ref_off, ref_len, query_off, query_len} into codes, out[k] = {score,
 * ref_end, read_end, ref_begin, read_end - read_begin, lanes used (16 / 8)} - all zero (lanes 16) for an empty operand or score 0.
 * where = CTO_REALIGN_DEVICE: the k_sw launches on `stream`; CTO_REALIGN_HOST: the SSE2 passes on host_threads workers.  Same numbers. */
int cto_sw_ends_batch(int n, const int8_t* codes, size_t n_codes, const int32_t* desc, int where, int host_threads, void* stream, int32_t* out);
/* cto_ssw_align for n independent alignments (Aligner::Align, ssw_cpp.cpp:78-215, as the realigner calls it): the striped passes, the
 * banded traceback (ssw.c:531-741) and the CIGAR over {S = X I D}.  codes / desc as for cto_sw_ends_batch; score[k] = 0 and an empty
 * CIGAR where the reference returns no alignment.  CIGAR text as cto_realign_reads writes it: NUL-terminated one after the other in
 * cigar_buf, cigar_off[n + 1] offsets.  where = CTO_REALIGN_DEVICE: k_sw + k_banded on `stream` (tracebacks with a first band over
 * 1 024 or more than 29 runs are done on the host inside the call); CTO_REALIGN_HOST: host_threads workers.  Same bytes. */
int cto_ssw_align_batch(int n, const int8_t* codes, size_t n_codes, const int32_t* desc, int where, int host_threads, void* stream,
                        int32_t* score, int32_t* ref_begin, char* cigar_buf, size_t cigar_cap, int64_t* cigar_off);
/* One striped Smith-Waterman pass alone (ssw.c:118-311 / :341-529 of the reference's src/realign: sw_sse2_byte / sw_sse2_word), test
 * hook: ref / read are base codes 0..4, lanes 16 (bytes) or 8 (words), out[4] = {score (255 on 8-bit overflow), ref_end, read_end,
 * overflow}. */
int cto_ssw_pass(const int8_t* ref, int ref_len, int reverse, const int8_t* read, int read_len, int lanes, int terminate, int32_t* out);
/* Worker threads of the Smith-Waterman stage inside cto_realign_reads (reads no haplotype took with <= 2 mismatches): default 1 (the
 * reference runs one single-threaded process per chunk), environment CTO_REALIGN_THREADS; output independent of the count. */
int cto_set_realign_threads(int n);
/* Reference positions one read contradicts (src/realign_reads.py:306-352: mismatches with BQ >= min_bq on A/C/G/T reference bases,
 * [p - n, p + n) around clean insertions / soft clips, deleted positions; the last two only inside [lo_ok, hi_ok]): one entry of out per
 * increment of the evidence counter.  bq = SAM quality text, positions 0-based, ref covers [ref0, ref0 + ref_len).  Returns the count,
 * or -1 when the Python loop it replaces would raise (index out of range) or cap is too small - the caller then runs that loop. */
int64_t cto_realign_read_evidence(const char* seq, int64_t seq_len, const char* bq, int64_t bq_len, const char* cigar, int64_t start,
                                  const char* ref, int64_t ref_len, int64_t ref0, int64_t lo_ok, int64_t hi_ok, int min_bq,
                                  int32_t* out, int64_t cap);
int cto_dbg_consensus(const char* ref, int n_reads, const char* const* reads, const int32_t* lowbq, const int64_t* lowbq_off,
                      char* buf, size_t cap, size_t* used);

#ifdef __cplusplus
}
#endif
#endif /* CLAIRSTO_AMD_H */
