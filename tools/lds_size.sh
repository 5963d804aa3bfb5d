#!/bin/bash
# LDS bytes per env of the physics kernels for a few body capacities (bp_lds_map in csrc/bp_device.hpp compiled for the host)
cd "$(dirname "$0")/.."
(printf '#include <cstdio>\n#include <initializer_list>\n#define __host__\n#define __device__\n'; grep -E "^#define BP_(QCAP|EVCAP|MBOX|NSLOT|PROFN) " benchpush_amd/csrc/bp_device.hpp; sed -n '/^struct LdsMap {/,/^};/p' benchpush_amd/csrc/bp_device.hpp; sed -n '/inline LdsMap bp_lds_map/,/^}/p' benchpush_amd/csrc/bp_device.hpp; echo 'int main(){ for (int nb : {64, 264, 272, 450, 520}) { printf("nbcap %d: box %u B, ship %u B (8 workgroups per CU need <= 20480)\n", nb, bp_lds_map(nb, nb>192?nb:192, true, false).total, bp_lds_map(nb, nb>192?nb:192,false,false).total);} }') > /tmp/lds.cpp && g++ -o /tmp/lds /tmp/lds.cpp && /tmp/lds
