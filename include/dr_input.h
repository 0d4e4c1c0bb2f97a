/* dr_input.h -- host-side input boundary of the hot path: TFRecord files of tf.Example
 * records, as written by the reference's deep_recommenders/datasets/movielens.py:54-92 (`_serialize_example`,
 * `serialize_tfrecords`) and parsed by `MovieLens.dataset` (:116-131, tf.io.parse_example with FixedLenFeature([], int64),
 * FixedLenFeature([], string), VarLenFeature(string)).
 *
 * Plain C ABI, host memory only, no TensorFlow / protobuf dependency.  The outputs are the layouts the device kernels of
 * dr_hotpath.h consume: int64 arrays (dr_hash_bucket_i64 / dr_vocab_lookup_i64) and byte blobs with offsets
 * (dr_hash_bucket_bytes / dr_vocab_lookup_bytes); a VarLenFeature additionally yields CSR row splits (the bag layout of
 * dr_emb_pool_fwd).
 *
 * Wire formats restated here (neither lives in /root/reference; both are TensorFlow's public on-disk formats):
 *   TFRecord  : { uint64 length (LE) | uint32 masked_crc32c(length) | byte data[length] | uint32 masked_crc32c(data) }*
 *               masked_crc(c) = ((c >> 15) | (c << 17)) + 0xa282ead8, CRC-32C (Castagnoli, reflected 0x82F63B78)
 *   tf.Example: protobuf  Example{Features features=1}  Features{map<string,Feature> feature=1}
 *               Feature{oneof: BytesList bytes_list=1 | FloatList float_list=2 | Int64List int64_list=3}
 *               BytesList{repeated bytes value=1}  FloatList{repeated float value=1 [packed]}  Int64List{repeated int64 value=1 [packed]}
 */
#ifndef DR_INPUT_H_
#define DR_INPUT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRI_OK 0
#define DRI_EINVAL (-1)     /* bad argument */
#define DRI_EIO (-10)       /* cannot open / short read */
#define DRI_ECORRUPT (-11)  /* TFRecord framing or CRC mismatch */
#define DRI_EPARSE (-12)    /* malformed protobuf, or a FixedLenFeature([]) key missing / wrong kind / not exactly one value */
#define DRI_ECAPACITY (-13) /* an output array is too small (sizes are reported so the caller can retry) */

const char* dri_version(void);

/* CRC-32C of a buffer (exposed for known-answer tests: crc32c("123456789") == 0xE3069283). */
uint32_t dri_crc32c(const uint8_t* data, int64_t n);

/* Scan a TFRecord file: payload byte offset and length of every record (tf.data.TFRecordDataset, movielens.py:127).
 * count_out receives the number of records in the file even when it exceeds `capacity` (then only the first
 * `capacity` entries are written and DRI_ECAPACITY is returned).  verify_crc != 0 checks both CRCs of every record. */
int dri_tfrecord_index(const char* path, int32_t verify_crc, int64_t* offsets, int64_t* lengths, int64_t capacity,
                       int64_t* count_out);

/* Read n record payloads (from dri_tfrecord_index) back to back into `out`; rec_offsets[n+1] receives their start
 * positions inside `out` (rec_offsets[n] = total bytes). */
int dri_tfrecord_read(const char* path, const int64_t* offsets, const int64_t* lengths, int64_t n, uint8_t* out,
                      int64_t out_capacity, int64_t* rec_offsets);

/* tf.io.FixedLenFeature([], tf.int64) of `key` for n serialized Examples -> out[n]. */
int dri_example_int64(const uint8_t* records, const int64_t* rec_offsets, int64_t n, const char* key, int64_t* out);

/* Bytes feature `key` of n serialized Examples.
 *   varlen == 0: tf.io.FixedLenFeature([], tf.string) -- exactly one value per example (else DRI_EPARSE)
 *   varlen != 0: tf.io.VarLenFeature(tf.string)      -- any number of values; a missing key is an empty row
 * totals_out[0] = number of values, totals_out[1] = number of bytes (always written).  With blob == NULL only the totals
 * are computed (size query).  Otherwise: value bytes back to back in `blob`, value_offsets[values+1] their positions,
 * row_splits[n+1] the first value of every example (CSR; row_splits[i+1]-row_splits[i] == 1 when varlen == 0). */
int dri_example_bytes(const uint8_t* records, const int64_t* rec_offsets, int64_t n, const char* key, int32_t varlen,
                      uint8_t* blob, int64_t blob_capacity, int64_t* value_offsets, int64_t values_capacity,
                      int64_t* row_splits, int64_t* totals_out);

#ifdef __cplusplus
}
#endif
#endif /* DR_INPUT_H_ */
