#!/bin/bash
# collect_slice_dbg.sh <tag> -- run ON THE GPU BOX (gpurun), after `make -C polychordlite_amd/csrc ../libpolychord_hip_slicedbg.so`
# here: cycles of the sections of a slice inside k_slice (chain 0 of every nursery, s_memtime) at the metric configuration,
# and the single-wave latencies of tools/ubench.hip; summary to gpurun_out/<tag>_slice_cycles.json (copy into profiles/).
set -u
tag=${1:-r03}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
hipcc -O2 --offload-arch=gfx950 tools/ubench.hip -o /tmp/ubench 2>/dev/null && /tmp/ubench | tail -2 > gpurun_out/${tag}_ubench.txt
PCHIP_LIB=$PWD/polychordlite_amd/libpolychord_hip_slicedbg.so PC_DEBUG=4 python tools/dev/gpu_slice_dbg.py gpurun_out/${tag}_ubench.txt > gpurun_out/${tag}_slice_cycles.json 2> gpurun_out/${tag}_slice_dbg.err
# (an empty file here = the SLICE_DBG library is older than the sources: `make -C polychordlite_amd/csrc ../libpolychord_hip_slicedbg.so` first)
[ -s gpurun_out/${tag}_slice_cycles.json ] || { rm -f gpurun_out/${tag}_slice_cycles.json; echo "collect_slice_dbg.sh: no section counters (see gpurun_out/${tag}_slice_dbg.err)" >&2; exit 1; }
cat gpurun_out/${tag}_slice_cycles.json
