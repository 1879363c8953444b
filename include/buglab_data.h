/* libbuglab_data -- native reader for BugLab's `*.msgpack.l.gz` shards.
 *
 * Replaces, for the graph part of a datapoint, the host-side decode chain of the reference:
 *   gzip + msgpack.Unpacker + OrderedDict construction   buglab/utils/msgpackutils.py:11-14
 *   add_open_vocab_nodes_and_edges                        buglab/representations/data.py:97-121
 *   _as_np_array over every edge list                     buglab/representations/data.py:124-127, 152-155
 *   subtoken splitting + vocabulary lookup of every node  ptgnn StrElementRepresentationModel.tensorize,
 *                                                         configured at buglab/models/modelregistry.py:59-82
 * Plain C ABI, host only (no HIP): the caller copies what it needs before the next call on the same reader.
 * Everything that is NOT graph.nodes / graph.edges / graph.reference_nodes is handed back as one msgpack map
 * (`rest`) for the Python side to decode -- those fields are small.  Re-entrant: one reader per thread. */
#ifndef BUGLAB_DATA_H
#define BUGLAB_DATA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct bl_reader bl_reader;
typedef struct bl_vocab bl_vocab;

typedef struct {
  int32_t is_nil;                    /* the stream element was msgpack nil (the reference skips those) */
  int32_t num_nodes;                 /* including the subtoken nodes appended for HasSubtoken edges */
  int32_t num_file_nodes;            /* nodes present in the file (the first num_file_nodes strings) */
  int32_t created_has_subtoken;      /* 1: the last edge kind, "HasSubtoken", was created by this reader */
  const char* node_text;             /* node strings, UTF-8, concatenated */
  const int32_t* node_text_off;      /* [num_nodes + 1] byte offsets into node_text */
  int32_t num_edge_kinds;
  const char* const* edge_kind;      /* NUL-terminated names, file order; "HasSubtoken" appended if created here */
  const int32_t* const* edge_pairs;  /* per kind: int32 [count][2] (source, target) */
  const int32_t* const* edge_feat;   /* per kind: int32 [count], index into feat_text or -1 (edge has no 3rd element) */
  const int32_t* edge_count;
  const char* feat_text;             /* edge feature strings (3rd element of an edge), concatenated */
  const int32_t* feat_text_off;      /* [num_feats + 1] */
  int32_t num_feats;
  const int32_t* reference_nodes;
  int32_t num_reference_nodes;
  int32_t non_ascii_identifier;      /* 1: an identifier token has non-ASCII bytes -- its subtokens were NOT
                                        created here (Unicode lower-casing is left to the Python path) */
  const uint8_t* rest;               /* msgpack map: every other top-level key; "graph" -> map of its other keys */
  int64_t rest_len;
} bl_datapoint_t;

const char* bl_data_last_error(void);
int32_t bl_data_version(void);

bl_reader* bl_reader_open(const char* path);                 /* NULL on error */
/* 1 = `out` filled, 0 = end of stream, < 0 = malformed input (message in bl_data_last_error()) */
int32_t bl_reader_next(bl_reader* r, bl_datapoint_t* out);
void bl_reader_close(bl_reader* r);

/* Iteration order of a CPython (3.7 - 3.12) `set` after inserting the non-negative ints inserted[0..n) in that order; out
 * has room for n values, the return value is the number of distinct keys (< 0: bad argument).  The reference walks the set
 * of NextToken endpoints in this order when it creates the subtoken nodes (buglab/representations/data.py:98-109), so the
 * order is part of the data contract; the reader uses it internally, this entry point exists for the tests. */
int32_t bl_pyset_order(const int32_t* inserted, int32_t n, int32_t* out);

/* vocabulary of subtokens: token i = text[off[i] .. off[i+1]) */
bl_vocab* bl_vocab_create(const char* text, const int32_t* off, int32_t n);
void bl_vocab_free(bl_vocab* v);
/* ids[n][S] (zero-padded) and lens[n] = max(1, #subtokens kept) for node strings; subtokens = snake_case /
 * camelCase parts, lower-cased, at most S.  needs_python[i] = 1 for strings with non-ASCII bytes (left zero). */
int32_t bl_tensorize_nodes(const bl_vocab* v, int32_t unk_id, const char* text, const int32_t* off, int32_t n, int32_t S,
                           int32_t* ids, int32_t* lens, uint8_t* needs_python);

/* ---- native collator of the graph part of a minibatch (buglab/data/collate.py::collate_graphs) ----
 * replaces GnnBugLabModel.initialize/extend/finalize_minibatch's Python list appends, gnn.py:431-604. */
typedef struct {
  int32_t num_nodes;
  int32_t token_stride;           /* columns of token_ids (<= S of the minibatch) */
  const int32_t* token_ids;       /* [num_nodes][token_stride] */
  const int32_t* token_lens;      /* [num_nodes] */
  const int32_t* const* adj;      /* T pointers to int32 [count][2] (source, target), graph-local node ids */
  const int32_t* adj_count;       /* [T] */
  const int32_t* const* adj_feat; /* optional (NULL = none): T pointers to int32 [count], one value per edge -- the edge-feature
                                   * token ids of a model with edge_feature_size > 0 (modelregistry.py:70-86); they travel with
                                   * their messages into bl_collated_t.msg_feat.  (bl_data_version() >= 3) */
} bl_graph_in_t;

typedef struct {                  /* caller-allocated outputs; N = sum of num_nodes, E = sum of all adj_count */
  int64_t num_nodes, num_messages;
  int32_t* token_ids;             /* [N][S] */
  int32_t* token_lens;            /* [N] */
  int32_t* msg_src; int32_t* msg_tgt;   /* [E] type-major, target-sorted inside a type */
  int32_t* type_ptr;              /* [T + 1] */
  int32_t* tgt_ptr; int32_t* tgt_msgs;  /* [N + 1], [E] */
  int32_t* src_ptr; int32_t* src_msgs;  /* [N + 1], [E] */
  int32_t* node_order;            /* [N] nodes with more than hub_degree incident messages first (by degree), then the rest */
  int64_t occ_capacity;           /* >= number of valid subtoken slots; sizes tok_occ, tok_chunk_id, tok_chunk_ptr (+1) */
  int32_t* tok_occ; int32_t* tok_chunk_ptr; int32_t* tok_chunk_id;
  int64_t num_occ, num_chunks;    /* written by the call */
  int32_t* msg_feat;              /* optional (NULL = none): [E] the graphs' adj_feat values in message order; every graph must
                                   * then supply adj_feat.  (bl_data_version() >= 3) */
} bl_collated_t;

int32_t bl_collate_graphs(const bl_graph_in_t* graphs, int32_t B, int32_t T, int32_t S, int32_t hub_degree,
                          int32_t token_chunk, bl_collated_t* out);

/* Stable counting sort of keys in [0, K): perm[E] = item indices ordered by key (ties keep input order),
 * ptr[K + 1] = where each key's run starts.  The collator's CSRs and per-type target order
 * (reference gnn.py:463-542 builds these with Python lists; here: one pass). */
int32_t bl_counting_sort(const int32_t* keys, int64_t E, int32_t K, int32_t* ptr, int32_t* perm);

#ifdef __cplusplus
}
#endif
#endif
