cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out /tmp/ub
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -Wno-unused-value -o /tmp/ub/gather_mix tools/ubench/gather_mix.hip > gpurun_out/r06_gather_mix_build.log 2>&1
/tmp/ub/gather_mix > gpurun_out/r06_gather_mix6.txt 2>&1
cat gpurun_out/r06_gather_mix6.txt
