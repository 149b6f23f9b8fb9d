#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/err
# (tools/debug/build_pk_variant.sh builds libmp_pk_all.so)
for lib in libmobileposer_hip.so libmp_pk_all.so; do
  for mode in 3 1; do
    MP_LIB_PATH=$PWD/mobileposer_amd/$lib timeout 300 python tools/debug/erratum.py $mode 20 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/err/probe.log
  done
done
