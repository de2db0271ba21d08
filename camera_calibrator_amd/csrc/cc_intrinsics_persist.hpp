// cc_intrinsics_persist.hpp -- interface between the host side of the intrinsics solver (cc_intrinsics.hip) and the
// persistent per-solve kernel (cc_intrinsics_persist.hip).
#pragma once
#include "cc_intrinsics_dev.hpp"

namespace cc {

constexpr int kPMaxTeams = 4;                     // frames (teams of four waves) per worker workgroup: 1, 2 or 4
constexpr int kPStatCols = 16;                    // doubles per statistics row (4 sums; + 9 diagonal sums in the first round)
constexpr int kPBcastWords = 64;                  // words per broadcast box
constexpr int kPMaxWorkers = 256;                 // worker workgroups: the control polls one statistics row per four of its 1024 threads (gather_stats4)
constexpr int kPLeaderRows = 16;                  // elimination rows a leader workgroup adds up before the control sees them

// Seam mailboxes of one handle: self-validating 8-byte words {epoch32 : half of a double}, agent-scope stores and
// loads (cc_intrinsics_persist.hip). Device memory, zeroed at creation; epochs only ever grow.
struct PersistDev {
  unsigned long long* sbox;   // [G][2 * kPStatCols]   worker g's statistics row
  unsigned long long* pbox;   // [G][2 * kPartialCols] worker g's elimination row (under the assumed decision)
  unsigned long long* rbox;   // [G][2 * kPartialCols] ... of its second elimination in a round (after a decision the workers did not assume)
  unsigned long long* lbox;   // [ceil(G / 16)][2 * kPartialCols] sum of sixteen elimination rows (by the leader among them)
  unsigned long long* xbox;   // [kPBcastWords] the control's broadcasts: flags, radius, nine doubles (step; Jacobi scales in the first round)
  unsigned* fail;             // [1] a wait inside the kernel timed out
  int32_t G;                  // worker workgroups; the control workgroup is block G
  int32_t teams;              // frames per worker workgroup (1, 2, 4): workgroups of 256 * teams threads
  uint32_t epoch0;            // epochs of this launch: epoch0 + 1 ...
  int32_t max_rounds;         // hard bound of the round loop (max_iterations + 2); three epochs per round
  int32_t restart;            // start from the state of the last set_state (init arrays) instead of buffer 0
  int32_t timeout_shift;      // a wait gives up after 2^shift ticks of the 100 MHz wall clock: 27 (1.3 s) alone on the device,
                              // 30 (10.7 s, the mailbox exchange's patience) with peer ranks
  int32_t first_shift;        // ... and the waits of the FIRST round: 20 (10.5 ms) alone on the device -- a first round that has not
                              // come together by then (every workgroup started, one sweep of at most ~0.1 ms) never will, and the
                              // caller gets its rerun in the two-kernel form after 10 ms instead of 1.3 s; with peer ranks as above
};

// workgroups of the `teams`-frame variant that can be resident on `device` at once (occupancy query x CUs)
int persist_resident_workgroups(int device, int teams, int* out);
void persist_launch(const IntrDev& P, const PersistDev& Q, bool drop_control, hipStream_t stream);

}  // namespace cc
