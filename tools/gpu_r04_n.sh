#!/bin/bash
# Round-4 check N: where the backward blend and the sort end -- last working wave against the dispatch of the idle workgroups.
R=$GRAFT_REPO_ROOT; cd $R
EXA_RASTER_LIB=$R/exavatar_release_amd/_variants/probe_bwdline.so timeout 300 python tools/gpu_bwd_timeline.py 0 50 2>&1 | grep -v amdgpu

