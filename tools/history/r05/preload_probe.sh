#!/bin/bash
cd $GRAFT_REPO_ROOT/tools/probes
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/probe_plain kernarg_preload_probe.hip 2>/dev/null
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 -o /tmp/probe_preload kernarg_preload_probe.hip 2>/dev/null
for r in 1 2; do echo "== plain"; timeout 60 /tmp/probe_plain; echo "== preload"; timeout 60 /tmp/probe_preload; done
