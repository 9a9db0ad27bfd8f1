// knobs.h -- the development knobs of a context (included by common.h).
#pragma once

// Development knobs (speed experiments and test hooks; never results).  A context takes them from the environment
// ONCE, in smhip_init, and only when SMHIP_DEV=1 is set there too: a stray SMHIP_* variable in a user's environment
// changes nothing.  A live context's knobs can be set with smhip_dev_knob (refused unless the context was created
// under SMHIP_DEV=1): the tests' way of driving the rarely taken routes.
struct DevKnobs {
  bool enabled = false;        // SMHIP_DEV=1 at smhip_init
  int lane_blocks_per_cu = 0;  // SMHIP_LANE_BLOCKS_PER_CU: walker workgroups of 256 threads per CU
  int set_groups_per_cu = 0;   // SMHIP_SET_GROUPS_PER_CU: the set walk's 128-thread workgroups per CU, 1 .. 8 (0: twice the above)
  int xcd_ranges = 1;          // SMHIP_XCD_RANGES=0: one visiting-order range instead of eight
  int scan_threshold = 0;      // SMHIP_SCAN_THRESHOLD: blocked lanes that trigger a scan phase
  int classes = 0;             // SMHIP_CLASSES: cost classes of the visiting order
  int bucket_bits = 0;         // SMHIP_BUCKET_BITS: leaf-order resolution of the visiting order
  int handout = 0;             // SMHIP_HANDOUT: idle lanes of a wave that trigger a hand-out of queries
  int sdtw_ring = 0;           // SMHIP_SDTW_RING=1: ring sweep only
  int sync_run = 0;            // SMHIP_SYNC_RUN=1: smhip_map_run always reads its two totals back
  long long pool_min = -1;     // SMHIP_POOL_MIN: initial overflow-pool size (>= 0: also forces the reading route)
  int spec_tight = 0;          // SMHIP_SPEC_TIGHT: 1 halves the query capacity of a speculative batch, 2 the anchor capacity
  int float_gap = 0;           // SMHIP_FLOAT_GAP=1: division form of the chaining gap tests
  int classify_resolve = -1;   // SMHIP_CLASSIFY_RESOLVE=0: the classifier leaves every anchor with a passing predecessor to the DP kernels
  int classify_dense = -1;     // SMHIP_CLASSIFY_DENSE=0: lockstep predecessor scans for every tile of the chain classifier
  int sort_lines = -1;         // SMHIP_SORT_LINES: 0 = exact two-pass partition only; 2 / 4 = whole-line partition with tiles of 2048 / 4096 whatever the batch
  int walk_share = -1;         // SMHIP_WALK_SHARE: 0 = never share a query's walk between the lanes of a wave, 1 = always (default: batches up to 4 096 chunks)
  int walk_grab = 0;           // SMHIP_WALK_GRAB: shared walks, queries a wave takes at a time ...
  int walk_grab_upto = 0;      // SMHIP_WALK_GRAB_UPTO: ... in launches of up to this many queries per wave
  int second_search = 0;       // SMHIP_SECOND_SEARCH=1: queries beyond the hit cap are searched again in the reference's
                               // order (round 3) instead of ranking the set walk's hits
};
