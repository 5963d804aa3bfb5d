// Load-time launch policy of a ship-ice / maze handle, as a pure host function of (envs per GPU, wave slots of the device).
//
// Everything below used to be written in absolute env counts tuned on one 256-CU MI355X (pairing from 5120 envs, tight limits below 7168,
// chunks of 100 sub-steps above 8192).  What those numbers meant is "rounds of the device's wave slots": the step kernel holds two wavefronts
// per SIMD (bp_lds_map: 20 480 B of LDS and ~245 VGPRs per env), so a device has `slots = CUs x 4 SIMDs x 2 = 8 x CUs` of them (2 048 on an
// MI355X in SPX mode) and a batch of E envs is E / slots rounds.  A launch of up to ~2.5 rounds follows the chains of its heaviest envs, a
// longer one the sum of its work; on a partitioned device (CPX: 32 CUs) or another CU count the same regime
// boundaries sit at the same number of rounds, not at the same number of envs.
//
// No HIP here: tests/test_host_cpu.py queries the function through bp_launch_policy_query without a GPU.
#pragma once
#include <algorithm>

struct BpLaunchPolicy {
    int wave_slots;      // resident wavefronts of the step kernel the device holds: 8 per CU
    int pair_mode;       // 0: one env per wavefront; 2: two light envs per wavefront inside the scheduler
    int tight;           // pairing limits of the chain-bound regime (only envs that are light right now run paired)
    int pair_solo;       // positions of the dispatch order that start alone
    int chunk;           // sub-steps per scheduler chunk, 0 = no scheduler (plain one-wave-per-env launch)
    int pp_max_act, pp_max_work, pp_rate;   // when a half leaves its pair (pair_should_leave)
};

// rounds of the wave slots, in halves, that start each regime (MI355X: 2.5 rounds = 5 120 envs, 3.5 = 7 168, 4 = 8 192)
#define BP_POLICY_PAIR_FROM_HALF_ROUNDS 5
#define BP_POLICY_LOOSE_FROM_HALF_ROUNDS 7
#define BP_POLICY_SCHED_UPTO_HALF_ROUNDS 8

static inline BpLaunchPolicy bp_launch_policy(int num_envs, int num_cus, bool can_pair, bool maze)
{
    BpLaunchPolicy p{};
    p.wave_slots = std::max(1, num_cus) * 8;
    const long long half_rounds2 = 2LL * num_envs;   // compared against k x slots: E >= (k / 2) x slots  <=>  2 E >= k x slots
    const bool pairing = !maze && can_pair && half_rounds2 >= (long long)BP_POLICY_PAIR_FROM_HALF_ROUNDS * p.wave_slots;
    p.pair_mode = pairing ? 2 : 0;
    // chain-bound pairing regime: up to 3.5 rounds the launch is within a few per cent of its heaviest env's chain
    p.tight = pairing && half_rounds2 < (long long)BP_POLICY_LOOSE_FROM_HALF_ROUNDS * p.wave_slots ? 1 : 0;
    p.pair_solo = p.tight ? num_envs / 8 : 0;
    p.pp_max_act = p.tight ? 16 : 20;
    p.pp_max_work = p.tight ? 9 : 40;
    p.pp_rate = p.tight ? 70 : 200;
    // the preemptive scheduler pays while a launch is a few rounds of the slots (its turns even out the finish); above four rounds the tail is
    // amortised, and only pairing launches keep it (it is what lets an env leave its pair) with long chunks
    const bool few_rounds = half_rounds2 <= (long long)BP_POLICY_SCHED_UPTO_HALF_ROUNDS * p.wave_slots;
    p.chunk = few_rounds ? 40 : (pairing ? 100 : 0);
    return p;
}
