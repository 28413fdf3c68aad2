/*
 * povu_hip.h -- C ABI of the MI355X (gfx950) `decompose` hot path.
 *
 * This is the boundary a povu maintainer binds instead of the CPU calls that
 * sit between "GFA parsed" and "PVST ready to write" in
 *   povu::subcommands::decompose::do_decompose   app/subcommand/decompose.cpp:94-160
 * i.e. it replaces, for that path and nothing else:
 *   bd::VG (adjacency built by mto::from_gfa::to_bd)    src/mto/from_gfa.cpp:191-277
 *   bd::VG::componetize                                  src/povu/graph/bidirected.cpp:477-602
 *   pst::Tree::from_bd                                   src/povu/graph/spanning_tree.cpp:262-463
 *   povu::flubbles::find_flubbles                        src/povu/algorithms/flubbles.cpp:721-745
 * (INTEGRATION.md shows the reference-side stub.)
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * All functions return 0 / non-NULL on success; on failure they return
 * non-zero / NULL and write a message to `err` (when given).  There is no CPU
 * fallback: without a usable HIP device every compute entry point fails.
 *
 * Threading: a context owns one HIP stream and one workspace arena and must be
 * used by one thread at a time; independent contexts may be used concurrently
 * (the reference's FFI makes the same promise, povu-rs/src/graph.rs:425).
 */
#ifndef POVU_HIP_H
#define POVU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POVU_HIP_NIL 0xFFFFFFFFu
/* vertex sides, reference pgt::v_end_e (include/povu/graph/types.hpp:39-42) */
#define POVU_SIDE_L 0
#define POVU_SIDE_R 1
/* tip marks per vertex (reference bd::VG::tips_, src/mto/from_gfa.cpp:262-277) */
#define POVU_TIP_NONE 0
#define POVU_TIP_L 1
#define POVU_TIP_R 2

typedef struct povu_hip_ctx povu_hip_ctx;
typedef struct povu_hip_forest povu_hip_forest;

/* number of visible HIP devices (0 when none; never touches a device) */
int povu_hip_device_count(void);

/* create / destroy a context on `device` */
povu_hip_ctx *povu_hip_create(int device, char *err, size_t errlen);
void povu_hip_destroy(povu_hip_ctx *ctx);

/*
 * Row A (device part).  Copies the link arrays to HBM and builds the per-side
 * CSR of the bidirected graph there (what bd::VG::add_vertex/add_edge/add_tip
 * build on the CPU, bidirected.cpp:309-340).
 *   vid[n_vtx]   segment id of vertex idx (the loader adds vertices ascending by id)
 *   v1,v2[n_links] endpoint vertex idx, s1,s2[n_links] endpoint side (POVU_SIDE_*)
 *   tips[n_vtx]  POVU_TIP_* per vertex, or NULL to infer them as to_bd does
 * The graph stays resident until the next upload or povu_hip_destroy.
 */
int povu_hip_graph_upload(povu_hip_ctx *ctx, uint32_t n_vtx, const uint32_t *vid, uint32_t n_links,
			  const uint32_t *v1, const uint8_t *s1, const uint32_t *v2, const uint8_t *s2,
			  const uint8_t *tips, char *err, size_t errlen);

/* device time of the last upload by HIP events, milliseconds: [0] host-to-device copies, [1] CSR build (side
 * degrees, offsets, per-side sorted adjacency, other-end table, tips), [2] reverse-slot table */
int povu_hip_last_upload_times(const povu_hip_ctx *ctx, double out_ms[3]);

typedef struct {
	uint32_t rank;	/* this process' shard (component sharding, 0-based) */
	uint32_t world; /* number of shards; 0 or 1 = everything */
	uint32_t flags; /* POVU_HIP_F_* */
} povu_hip_opts;
#define POVU_HIP_F_HAIRPINS 1u /* also report hairpin boundaries (--hairpins, flubbles.cpp:712-717) */
#define POVU_HIP_F_SEQUENTIAL 2u /* force the one-lane-per-component kernels for every stage */
#define POVU_HIP_F_SEQ_TREE 4u /* sequential spanning tree, parallel classes/stack/PVST (A/B testing) */
#define POVU_HIP_F_FORCE_REDO 8u /* treat every component as flagged for the sequential redo (tests) */
#define POVU_HIP_F_REDO_ODD 256u /* treat every second component as flagged for the sequential redo (tests of the mixed result) */
#define POVU_HIP_F_NO_STAGE_TIMES 32u /* record only the pass total, not the per-stage HIP events */
#define POVU_HIP_F_BIG_CLASS_DFS 64u /* always walk the classes with the filtered-scan-list DFS large classes get (A/B testing) */
#define POVU_HIP_F_SPARSE_SPLITTERS 128u /* list ranking with 1-in-16 splitters instead of 1-in-8 (A/B testing) */
#define POVU_HIP_F_ALL_VERTEX_CLASSES 512u /* number the cycle classes of all tree edges, not just the black ones the candidate stack holds (A/B testing) */
#define POVU_HIP_F_CHECK_LAMINAR 1024u /* always run the laminarity check of the candidate stack's (prev, i) intervals; by default it only runs when the literal hi_2 rule capped differently from the second-highest reach, DESIGN.md section 4 has the proof for the other case (A/B testing, fuzzing) */
#define POVU_HIP_F_LEAF_SUBFLUBBLES 2048u /* the two relabelling passes of `-s`: find_tiny (tiny.cpp:100-129) and find_parallel (parallel.cpp:263-287) on every PVST; the forest then also carries ai / zi and the line letter of every vertex (povu_hip_forest_get_sub).  Not the reference's whole `-s`: that is POVU_HIP_F_SUBFLUBBLES */
#define POVU_HIP_F_SUBFLUBBLES 8192u /* all five passes of `-s` (app/subcommand/decompose.cpp:63-70): implies POVU_HIP_F_LEAF_SUBFLUBBLES, then find_concealed (concealed.cpp:1198-1243), find_midi (midi.cpp:225-268) and find_smothered (smothered.cpp:385-432) INSERT vertices into every PVST; the forest carries the extended trees (povu_hip_forest_get_subtree) and povu_hip_forest_pvst_text prints them.  Parity unpinned: the reference holds no T / O / C / M / S line; undefined behaviour of the reference is decided as oracle/povu_oracle_sub.inc lists */
#define POVU_HIP_F_ASYNC 4096u /* povu_hip_decompose returns as soon as the forest is laid out -- tree table, sizes, the page-locked result block -- while the last kernels and the copy of the PVST arrays to the host are still in flight; povu_hip_forest_wait (or any accessor of the forest: they wait by themselves) completes it.  A second povu_hip_decompose on the same context may start at once: its kernels run while the copy engine still moves the first result over PCIe.  Ignored (the call completes before it returns) without POVU_HIP_F_NO_STAGE_TIMES, with hairpins, subflubble labels, the test modes, and when the pass needs the laminarity check */
#define POVU_HIP_F_SORTED_ADJ 16u /* build the local adjacency with the radix sort hub graphs use (A/B testing) */

/*
 * Rows B-G.  Decomposes the resident graph: weakly connected components
 * (numbered 1.. by minimum vertex idx, decompose.cpp:129), component
 * re-indexing, biedged spanning tree, cycle-equivalence classes, candidate
 * stack, next_seen and the PVST of every component with >= 3 vertices
 * (decompose.cpp:135-142) that belongs to this shard.  The result lives in
 * host memory.
 */
povu_hip_forest *povu_hip_decompose(povu_hip_ctx *ctx, const povu_hip_opts *opts, char *err, size_t errlen);

/*
 * Row B on its own (what `povu info` and `povu prune` need, app/subcommand/info.cpp:19-45,
 * prune.cpp:17-41): the components exactly as bd::VG::componetize builds them -- ordered by minimum
 * vertex idx, vertices ascending, links in first-encounter order stored from the encountering side,
 * self loops as (ve, complement(ve)) (bidirected.cpp:552-569, :66-77).
 */
typedef struct {
	uint32_t n_components;
	uint32_t n_vtx, n_links;
	const uint32_t *vtx_off;  /* [n_components + 1] component c owns vertices [vtx_off[c], vtx_off[c+1]) */
	const uint32_t *link_off; /* [n_components + 1] */
	const uint32_t *vtx_id;	  /* [n_vtx] segment id, local vertex order, component-major */
	const uint8_t *vtx_tip;	  /* [n_vtx] POVU_TIP_* */
	const uint32_t *l_v1, *l_v2; /* [n_links] LOCAL vertex idx of the two ends, local link order */
	const uint8_t *l_s1, *l_s2;  /* [n_links] sides */
} povu_hip_components;
povu_hip_components *povu_hip_componetize(povu_hip_ctx *ctx, char *err, size_t errlen);
void povu_hip_components_free(povu_hip_components *c);

/* ---- multi-GPU: component sharding, one process per GPU ----
 * Replaces the static per-thread chunks of do_decompose (app/subcommand/decompose.cpp:78-92,116-157): the
 * components of the graph resident on the ROOT rank are labelled there (row B's union-find kernels), bin-packed
 * over the ranks (greedy longest-processing-time on links + segments), the links are partitioned on the device and
 * every rank receives the sub-graph of its components; ranks decompose independently; the PVST arrays are gathered
 * on the root.  No collective touches the traversal itself. */
typedef struct povu_hip_shards povu_hip_shards;
typedef struct {
	uint32_t n_vtx, n_links, n_components; /* of this shard */
	uint64_t weight;		       /* LPT load: links + segments (+1 per component) */
	size_t bytes;			       /* size of the packed shard */
	const void *device_ptr;		       /* packed shard in the root's HBM */
} povu_hip_shard_info;
/* LPT assignment of `n` weights to `world` ranks: heaviest first (stable), each to the least loaded rank (lowest
 * rank on ties), every placed item also costs 1.  Host only -- no GPU needed. */
int povu_hip_lpt_assign(const uint64_t *weights, uint32_t n, uint32_t world, uint32_t *owner_out);
/* Partitions the graph resident in `ctx` into `world` packed shards in device memory.  Vertices keep their
 * ascending global order inside a shard and links their L-line order, so a shard's own component numbering
 * preserves the global order. */
povu_hip_shards *povu_hip_shard_partition(povu_hip_ctx *ctx, uint32_t world, char *err, size_t errlen);
uint32_t povu_hip_shards_world(const povu_hip_shards *s);
uint32_t povu_hip_shards_total_components(const povu_hip_shards *s);
int povu_hip_shards_get(const povu_hip_shards *s, uint32_t rank, povu_hip_shard_info *out);
/* device time of the partition (HIP events, ms): [0] labelling, [1] weights + LPT, [2] partition kernels */
int povu_hip_shards_times(const povu_hip_shards *s, double out_ms[3]);
/* (the packed shards live in an arena of `ctx`: they stay valid until the next partition on that context or its destruction)
 * copies packed shard `rank` to host memory (`dst` holds info.bytes) -- for transports other than RCCL */
int povu_hip_shards_export(const povu_hip_shards *s, povu_hip_ctx *ctx, uint32_t rank, void *dst);
void povu_hip_shards_free(povu_hip_shards *s);
/* Makes a packed shard the resident graph of `ctx` (CSR built on the device).  `packed` is host memory
 * (on_device = 0) or memory of ctx's device (on_device = 1).  The shard's component ids (1-based ids in the whole
 * graph, ascending) are kept in the context for povu_hip_forest_globalize / the gather. */
int povu_hip_graph_upload_shard(povu_hip_ctx *ctx, const void *packed, size_t bytes, int on_device, char *err, size_t errlen);
/* number of components of the WHOLE graph the resident shard was cut from (0: the resident graph is no shard) */
uint32_t povu_hip_shard_total_components(const povu_hip_ctx *ctx);
/* rewrites the component ids of a forest computed on a shard to the ids of the whole graph */
int povu_hip_forest_globalize(povu_hip_forest *f, const povu_hip_ctx *ctx);
/* Wire format of a forest: [u64 n_trees, u64 total_entries, u64 total_components | per tree u32 component id, n_vtx,
 * n_links, n_pvst | a_id | z_id | parent (u32 x total) | a_or | z_or (u8 x total)], every section padded to 64 B. */
size_t povu_hip_forest_pack_size(const povu_hip_forest *f);
int povu_hip_forest_pack(const povu_hip_forest *f, void *dst, size_t cap);
/* merges packed forests (host memory) into one forest, trees ordered by component id */
povu_hip_forest *povu_hip_forest_merge(povu_hip_ctx *ctx, const void *const *packed, const size_t *bytes, uint32_t n,
				       char *err, size_t errlen);

/* RCCL communicator owned by the library (ncclSend / ncclRecv over xGMI on the context's stream).  The unique id
 * is created on one rank (povu_hip_comm_unique_id) and handed to the others by the launcher. */
typedef struct povu_hip_comm povu_hip_comm;
#define POVU_HIP_COMM_ID_BYTES 128
int povu_hip_comm_unique_id(char id[POVU_HIP_COMM_ID_BYTES], char *err, size_t errlen);
povu_hip_comm *povu_hip_comm_create(povu_hip_ctx *ctx, const char id[POVU_HIP_COMM_ID_BYTES], uint32_t rank, uint32_t world,
				    char *err, size_t errlen);
void povu_hip_comm_destroy(povu_hip_comm *c);
/* Scatter: on the root (rank 0) `shards` is the partition of its resident graph, elsewhere NULL.  On return every
 * rank's context holds its shard as resident graph (on the root it replaces the whole graph; a root that wants to
 * keep the whole graph resident scatters from a second context).  Collective: every rank of the communicator must
 * call it.  Two failures travel in the handshake and make the call fail on ALL ranks before any shard moves: the root has
 * no partition for exactly `world` ranks, a receiver has no room for its shard.  Bad ARGUMENTS (a null handle, a
 * destination context that is not the communicator's) fail on the calling rank alone, before its first collective: the
 * other ranks then wait -- a caller's bug, not a run-time condition.  EXPERIMENTAL: the library's own RCCL transfers have
 * never run on more than one GPU (the one-process engine below and the torch.distributed path of bench.py are what is
 * rehearsed).  Every RCCL call of a communicator runs on the communicator's own stream. */
int povu_hip_comm_scatter(povu_hip_comm *c, const povu_hip_shards *shards, povu_hip_ctx *dst_ctx, char *err, size_t errlen);
/* Gather: every rank passes the (globalized) forest of its shard; the root gets the merged forest, the others an
 * empty one.  Collective like the scatter: a forest that cannot travel (hairpin boundaries, a merged forest) or a root
 * without room fails the call on all ranks before any block moves. */
povu_hip_forest *povu_hip_comm_gather(povu_hip_comm *c, const povu_hip_forest *mine, char *err, size_t errlen);
/* wall time of the last scatter / gather on this rank, milliseconds */
int povu_hip_comm_times(const povu_hip_comm *c, double out_ms[2]);

/* ---- gather without a second PCIe crossing (several processes on one node) ----
 * A rank's decompose already lands its PVST block in page-locked HOST memory over its own GPU's PCIe link.  After
 * povu_hip_share_results the blocks of a context's forests are POSIX shared-memory segments ("/povu.<tag>.<k>",
 * page-locked and mapped for the device): the root maps a rank's segment by name and reads the arrays where they are.
 * What the ranks exchange is one 64-byte descriptor each (RCCL all-gather, torch.distributed, a pipe ...).
 * Replaces the per-thread ownership of do_decompose's workers (app/subcommand/decompose.cpp:116-157), which write from
 * their own memory. */
/* from now on this context's result blocks are shared segments named after `tag` (at most 96 characters, no '/');
 * the multi-process convention is tag = "<job>.<rank>" */
int povu_hip_share_results(povu_hip_ctx *ctx, const char *tag, char *err, size_t errlen);
/* describes the block of `f` for another process: desc = [magic, segment k, segment bytes, trees, entries, components of the
 * whole graph, offset of the tree table, 0 (the caller stores the sender's rank here)]; the tree table is written into
 * the segment behind the arrays.  A forest whose trees sit in several blocks is first brought into one.  Returns 2 when the
 * forest's block is no shared segment, 4 for hairpin boundaries / subflubble labels (they do not travel). */
int povu_hip_forest_share(povu_hip_forest *f, uint64_t desc[8]);
/* Root: the merged forest of `n` descriptors (8 words each, word 7 = sender rank; segments "/povu.<job_tag>.<rank>.<k>"
 * are mapped read-only and stay mapped in `ctx`) and of its own forest `own` (taken over, may be NULL; the descriptor
 * with rank `own_rank` is skipped).  The other ranks' arrays stay THEIR memory: the merged forest is valid until the
 * sender frees the forest it described -- in a collective loop, until the sender's next gather. */
povu_hip_forest *povu_hip_forest_attach(povu_hip_ctx *ctx, povu_hip_forest *own, uint32_t own_rank, const char *job_tag,
					const uint64_t *descs, uint32_t n, char *err, size_t errlen);
/* bytes this context has moved since it was created: [0] host to device, [1] device to host (copies and results kernels
 * write straight into page-locked memory), [2] sent to / [3] received from other GPUs (xGMI) */
int povu_hip_transfer_bytes(const povu_hip_ctx *ctx, uint64_t out[4]);

/* ---- one process, N GPUs (`povu decompose --gpus N`) ----
 * The multi-threaded form of do_decompose (app/subcommand/decompose.cpp:116-157: every worker owns its components from
 * graph to file) with GPUs for workers: one context and one host thread per device.  The root device labels the
 * components, bin-packs them (LPT) and partitions the links; the shards travel over xGMI (RCCL ncclSend / ncclRecv, one
 * communicator per device); every GPU decomposes its shard and copies its PVST block into page-locked host memory over
 * ITS OWN PCIe link -- the one address space makes that the gather; nothing returns through the root's link. */
typedef struct povu_hip_multi povu_hip_multi;
/* devices[n]: HIP device of every rank (rank 0 = root).  The same device may be named more than once (rehearsal on a
 * one-GPU box: shards are then loaded straight from the partition, no RCCL). */
povu_hip_multi *povu_hip_multi_create(const int *devices, uint32_t n, char *err, size_t errlen);
void povu_hip_multi_destroy(povu_hip_multi *m);
uint32_t povu_hip_multi_world(const povu_hip_multi *m);
/* the whole graph to the root device (povu_hip_graph_upload on the root's graph context) */
int povu_hip_multi_upload(povu_hip_multi *m, uint32_t n_vtx, const uint32_t *vid, uint32_t n_links, const uint32_t *v1,
			  const uint8_t *s1, const uint32_t *v2, const uint8_t *s2, const uint8_t *tips, char *err, size_t errlen);
/* label + LPT + partition on the root, shards to their devices, every rank builds its CSR.  With keep_graph = 0 the
 * root gives the whole graph's memory back first-thing after the partition (a CLI run never needs it again). */
int povu_hip_multi_scatter(povu_hip_multi *m, int keep_graph, char *err, size_t errlen);
/* Every rank decomposes its resident shard on its own thread (flags: POVU_HIP_F_*).  `sink`, when given, runs ON THE
 * WORKER'S THREAD with that rank's forest (global component ids) as soon as it is done -- a CLI formats and writes its
 * files there, like the reference's workers; a non-zero return fails the call.  Returns the merged forest (no array is
 * copied: it takes over every rank's blocks).  With POVU_HIP_F_ASYNC | POVU_HIP_F_NO_STAGE_TIMES in `flags` (and no sink) the
 * call returns while the ranks' PVST arrays are still on their way to the host -- the merged forest waits for them like any
 * POVU_HIP_F_ASYNC forest -- and the next call's kernels run under those copies. */
typedef int (*povu_hip_multi_sink)(uint32_t rank, const povu_hip_forest *f, void *user);
povu_hip_forest *povu_hip_multi_decompose(povu_hip_multi *m, uint32_t flags, povu_hip_multi_sink sink, void *user, char *err,
					  size_t errlen);
typedef struct {
	int device;
	uint32_t n_vtx, n_links, n_components; /* of the rank's shard */
	uint64_t shard_bytes;
	double recv_ms, csr_ms;		 /* last scatter: until the shard was there; CSR build */
	double decompose_ms, sink_ms;	 /* last decompose */
	uint64_t h2d, d2h, peer_out, peer_in; /* povu_hip_transfer_bytes of the rank's context */
} povu_hip_multi_rank_info;
int povu_hip_multi_rank(const povu_hip_multi *m, uint32_t rank, povu_hip_multi_rank_info *out);
/* [0] label, [1] weights + LPT, [2] partition kernels (device time on the root), [3] wall of the last scatter, [4] wall of
 * the last decompose (slowest rank + merge), [5] merge alone */
int povu_hip_multi_times(const povu_hip_multi *m, double out_ms[6]);
/* "none" (one rank), "rccl", "peer-copy" (hipMemcpyPeerAsync; also the fallback when RCCL cannot be initialised: the
 * reason is appended) or "same-device" */
const char *povu_hip_multi_transport(const povu_hip_multi *m);
/* the context of worker `rank` (NULL when out of range): for the debug exports of the pass povu_hip_multi_decompose just ran --
 * call them from the sink callback (it runs on the worker's own thread, the only one that may use the context then) */
povu_hip_ctx *povu_hip_multi_context(povu_hip_multi *m, uint32_t rank);
/* a context that holds a shard: ids[k] = id (1-based, of the whole graph) of the shard's k-th component, the rank the debug
 * exports and the sidecar of --structure-export address a component by.  0 on success, 1 when the resident graph is no shard. */
int povu_hip_shard_component_ids(const povu_hip_ctx *ctx, const uint32_t **ids, uint32_t *n);

/* Completes a forest of a POVU_HIP_F_ASYNC decompose (no-op otherwise): returns when its arrays are in host memory. */
int povu_hip_forest_wait(povu_hip_forest *f);
/* HIP-event time of the pass that produced `f`, from its first kernel to the last byte in host memory, milliseconds
 * (waits for the forest first; < 0 when the forest carries none: merged forests, empty shards) */
double povu_hip_forest_pass_ms(povu_hip_forest *f);
/* HIP-event time from the first kernel of the pass behind `first` to the last byte of the pass behind `last` (both of one
 * context): what a run of overlapped passes took on the device */
double povu_hip_forest_span_ms(povu_hip_forest *first, povu_hip_forest *last);

/* components of the WHOLE graph (all shards), including skipped ones */
uint32_t povu_hip_forest_total_components(const povu_hip_forest *f);
/* PVSTs held by this forest (this shard's components with >= 3 vertices) */
uint32_t povu_hip_forest_tree_count(const povu_hip_forest *f);

typedef struct {
	uint32_t component_id; /* 1-based, file name <id>.pvst */
	uint32_t n_vtx, n_links;
	uint32_t n_pvst; /* PVST vertices incl. the dummy root 0 */
	/* arrays of n_pvst entries, PVST vertex idx = emission order; entry 0 = dummy root */
	const uint32_t *a_id, *z_id;   /* flubble endpoints (segment ids) */
	const uint8_t *a_or, *z_or;    /* 0 forward '>', 1 reverse '<' */
	const uint32_t *parent;	       /* PVST parent idx, POVU_HIP_NIL for the root */
	uint32_t n_hairpins;	       /* with POVU_HIP_F_HAIRPINS */
	const uint64_t *hairpins;      /* pairs (b1,b2) */
} povu_hip_tree;

int povu_hip_forest_get(const povu_hip_forest *f, uint32_t i, povu_hip_tree *out);
/* With POVU_HIP_F_LEAF_SUBFLUBBLES: per PVST vertex of tree i (n_pvst entries, entry 0 = dummy root) the spanning-tree
 * vertices ai / zi that pvst::Flubble::create takes (compute_ai_zi, flubbles.cpp:264-290; POVU_HIP_NIL for the root) and
 * the line letter 'D' 'F' 'T' (tiny) 'O' (parallel).  Any of the three pointers may be NULL.  Returns 3 when the
 * forest carries none. */
int povu_hip_forest_get_sub(const povu_hip_forest *f, uint32_t i, const uint32_t **ai, const uint32_t **zi, const uint8_t **fam);
/* With POVU_HIP_F_SUBFLUBBLES: tree i after all five passes of -s (pvst::Tree of include/povu/graph/pvst.hpp:719-900 as
 * write_pvst sees it).  Vertices [0, n_flubble_like) are those of povu_hip_forest_get, relabelled; the inserted ones follow
 * in the order the reference adds them (concealed, midi, smothered).  Per vertex: the line letter ('D' 'F' 'T' 'O' 'C' 'M'
 * 'S'), the two boundaries in the order as_str() prints them (orientation 0 = '>'), the route letter ('L' 'R', 0 = none) and
 * its children, in the reference's order: child[child_off[v] .. child_off[v + 1]).  Returns 3 when the forest carries none. */
typedef struct {
	uint32_t n_total, n_flubble_like, n_concealed, n_midi, n_smothered;
	const uint8_t *fam, *or1, *or2, *route;
	const uint32_t *id1, *id2;
	const uint32_t *child_off; /* [n_total + 1], offsets into `child` */
	const uint32_t *child;
} povu_hip_subtree;
int povu_hip_forest_get_subtree(const povu_hip_forest *f, uint32_t i, povu_hip_subtree *out);
/* write_pvst (src/mto/to_pvst.cpp:31-109) of such a tree; malloc'd, free with povu_hip_buffer_free */
char *povu_hip_pvst_format_subtree(const povu_hip_subtree *t, size_t *len);
/*
 * The forest's arrays as ONE page-locked host block (what a multi-GPU gather ships):
 * offsets[0..4] = byte offsets of a_id, z_id, parent (u32 x total) and a_or, z_or (u8 x total);
 * tree i occupies entries [first[i], first[i] + n_pvst) of every array (first = povu_hip_forest_first).
 */
int povu_hip_forest_raw(const povu_hip_forest *f, const void **block, size_t *bytes, uint64_t *total, uint64_t offsets[5]);
uint64_t povu_hip_forest_first(const povu_hip_forest *f, uint32_t i);
void povu_hip_forest_free(povu_hip_forest *f);

/*
 * Serialises tree `i` exactly as mto::to_pvst::write_pvst does
 * (src/mto/to_pvst.cpp:23-109).  Returns a malloc'd buffer (free with
 * povu_hip_buffer_free) and its length.
 */
char *povu_hip_forest_pvst_text(const povu_hip_forest *f, uint32_t i, size_t *len);
/* the same serialiser on caller-provided PVST arrays (host only, no GPU needed); NULL on bad input */
char *povu_hip_pvst_format(uint32_t n_pvst, const uint32_t *a_id, const uint32_t *z_id, const uint8_t *a_or,
			   const uint8_t *z_or, const uint32_t *parent, size_t *len);
/* ... with the line letter of every vertex given ('D' for entry 0, then 'F' / 'T' / 'O'); fam == NULL = all flubbles */
char *povu_hip_pvst_format_fam(uint32_t n_pvst, const uint32_t *a_id, const uint32_t *z_id, const uint8_t *a_or,
			       const uint8_t *z_or, const uint32_t *parent, const uint8_t *fam, size_t *len);
void povu_hip_buffer_free(void *p);

/* ---- PVST reader (host only): mto::from_pvst::read_pvst + pvst::Tree::comp_heights,
 * src/mto/from_pvst.cpp:162-302, include/povu/graph/pvst.hpp:807-836 ---- */
typedef struct {
	uint32_t n;	    /* vertices in file order */
	char *type;	    /* D F T O M C S */
	uint32_t *file_id;  /* second column */
	uint32_t *a_id, *z_id;
	uint8_t *a_or, *z_or;
	uint8_t *route;	    /* 0 = L (start to end), 1 = R */
	uint32_t *parent;   /* idx of the parent vertex, POVU_HIP_NIL for the root */
	uint32_t *height;   /* distance from the root */
} povu_pvst_doc;
povu_pvst_doc *povu_pvst_parse(const char *text, size_t len, char *err, size_t errlen);

/* GFA v1 text of a whole graph in the arrays povu_hip_graph_upload takes (mto::to_gfa::write_gfa's record shapes,
 * src/mto/to_gfa.cpp:13-56): `S <id> A` per segment, `L <a> <+|-> <b> <+|-> 0M` per link, `+` = out of a's r side / into
 * b's l side -- the inverse of the loader contract, so that writing and loading a graph is the identity.  Host only. */
int povu_hip_gfa_write(const char *path, uint32_t n_vtx, const uint32_t *vid, uint32_t n_links, const uint32_t *v1,
		       const uint8_t *s1, const uint32_t *v2, const uint8_t *s2, char *err, size_t errlen);
void povu_pvst_doc_free(povu_pvst_doc *doc);

/* ---- measurement (bench.py, povu-stage-cost lines) ---- */
typedef struct {
	char name[48];	  /* kernel group */
	double ms;	  /* HIP-event time on the context's stream, last decompose */
	uint32_t launches;
} povu_hip_stage_time;
/* stage timings of the last povu_hip_decompose on this context */
int povu_hip_last_stage_times(const povu_hip_ctx *ctx, povu_hip_stage_time *out, int max);
/* components redone by the one-lane kernels in the last decompose (only the test modes POVU_HIP_F_FORCE_REDO / _REDO_ODD
 * send any: crossing candidate-stack intervals, the one case that used to, are resolved by the parallel stage itself) */
uint32_t povu_hip_last_seq_redo(const povu_hip_ctx *ctx);
/* 1 when the last decompose numbered the cycle classes of the black tree edges only (the default whenever the
 * literal hi_2 rule of flubbles.cpp:566-574 picked the second-highest reach everywhere and no hairpins were asked
 * for), 0 when it went over all tree edges */
int povu_hip_last_black_only_classes(const povu_hip_ctx *ctx);
/* 1 when the last decompose ran the laminarity check of the candidate stack's (prev, i) intervals (only when the literal
 * hi_2 rule capped differently from the second-highest reach somewhere, or with POVU_HIP_F_CHECK_LAMINAR) */
int povu_hip_last_laminar_check_ran(const povu_hip_ctx *ctx);
/* after a pass that ran the laminarity check: out[0] = candidate-stack entries whose (previous occurrence, this occurrence)
 * interval holds an entry that reaches back beyond it, out[1] = those whose class had really been popped off
 * add_flubbles' stack by then (flubbles.cpp:326-343) -- decided in place by the parallel stage, no sequential redo */
int povu_hip_last_crossings(povu_hip_ctx *ctx, uint32_t out[2]);
/* number of links in the components this shard processed in the last decompose */
uint64_t povu_hip_last_links_processed(const povu_hip_ctx *ctx);

/* ---- stage-level parity hooks (tests only; device state of the last decompose) ----
 * After a pass that redid SOME components with the one-lane kernels (povu_hip_last_seq_redo() between 1 and the
 * component count - 1) the classes and candidate stacks sit in two layouts: povu_hip_debug_tree (when `cls` is asked
 * for), povu_hip_debug_edge_ids and povu_hip_debug_stack then return 4 instead of exporting half-valid state. */
/* copies comp_of[v] (0-based component rank) and local vertex idx for every GLOBAL vertex idx */
int povu_hip_debug_components(povu_hip_ctx *ctx, uint32_t *comp_of, uint32_t *local_idx);
/* tree arrays of component `comp` (0-based rank): sizes via n_tree first call with NULLs; `cls` is defined for the
 * child ends of black tree edges, and for the others too only when povu_hip_last_black_only_classes() == 0 */
int povu_hip_debug_tree(povu_hip_ctx *ctx, uint32_t comp, uint32_t *n_tree, uint32_t *gid, uint8_t *typ,
			uint32_t *par, uint32_t *cls);
/* id of the tree edge into each tree vertex (tree_edge_id[0] = 0xFFFFFFFF): tree and back edges share one counter in
 * creation order (Tree::add_tree_edge / add_be, spanning_tree.cpp:784-805).  Conformance export only; returns 3 after a
 * pass that built the tree with the one-lane kernels (POVU_HIP_F_SEQUENTIAL / POVU_HIP_F_SEQ_TREE). */
int povu_hip_debug_edge_ids(povu_hip_ctx *ctx, uint32_t comp, uint32_t *n_tree, uint32_t *tree_edge_id);
/* candidate stack of component `comp`: tree vertex of each entry, class, next_seen */
int povu_hip_debug_stack(povu_hip_ctx *ctx, uint32_t comp, uint32_t *n, uint32_t *tree_vtx, uint32_t *cls,
			 uint32_t *next_seen);

/* unit-test hook for the device-wide scans of the path: exclusive scan of in[0..n) (op 0 = sum mod 2^32,
 * 1 = running maximum) and, when in2 is given, an independent sum scan of in2[0..n2) in the same launch */
int povu_hip_debug_scan(povu_hip_ctx *ctx, int op, const uint32_t *in, uint32_t *out, size_t n, const uint32_t *in2,
			uint32_t *out2, size_t n2);
/* timing hook (tools/scan_time.py): `reps` exclusive scans (op as above) of n device-resident words, ms per scan by HIP
 * events; < 0 on error */
double povu_hip_debug_scan_time(povu_hip_ctx *ctx, size_t n, int reps, int op);

/* device workspace (bytes) one plain povu_hip_decompose call reserves for a graph of this size whose vertices come grouped
 * by component and that has no hub vertex (> 48 links) and no self loop -- a pangenome GFA --, excluding the resident graph
 * itself (~42 B/link + 13 B/segment), the one-lane kernels' lists and the wave walk's arrays (44 B per side, taken by the
 * first pass that meets a 2-edge-connected class of more than 256 sides); sides without links are assumed to number two per
 * component and one segment in 64.  The general case: povu_hip_workspace_breakdown.  n_components = 0 assumes the worst
 * case (every segment its own component); 0 when it cannot be computed */
uint64_t povu_hip_workspace_estimate(uint32_t n_vtx, uint32_t n_links, uint32_t n_components);
/* the parts of that estimate: [0] rows A/B state of a graph whose vertices come grouped by component, without hub vertices
 * or self loops ([6]: the general case), [1] tree / result arrays all execution modes share, [2] / [3] class stage: arrays
 * the tree stage already writes / arrays first written by the class stage, [4] / [5] tree stage: arrays that outlive it / its
 * own.  [3] and [5] share one stretch (the larger of the two counts) */
int povu_hip_workspace_breakdown(uint32_t n_vtx, uint32_t n_links, uint32_t n_components, uint64_t out[7]);
/* Reserves the device memory a graph of this size will need (resident graph, CSR build scratch, decompose workspace for
 * the worst case of components) on a context that holds nothing yet, so that the first upload + decompose do not pay for
 * the allocation: meant to run on another thread while the caller still parses its input, when that takes longer than
 * bringing the runtime up (the CLI does so with POVU_CLI_PREWARM=1; exact sizes, no head room).  Best effort:
 * returns 0 and reserves nothing when the worst case does not fit; 1 + message only for bad arguments / HIP errors.  Must
 * not run concurrently with another call on the same context. */
int povu_hip_prewarm(povu_hip_ctx *ctx, uint32_t n_vtx, uint32_t n_links, char *err, size_t errlen);
/* upper bound of the EXTRA device memory POVU_HIP_F_LEAF_SUBFLUBBLES takes while its stage runs (allocated and released
 * inside the call; n_components = 0: not known) */
uint64_t povu_hip_leaf_workspace_estimate(uint32_t n_vtx, uint32_t n_components);

const char *povu_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif
