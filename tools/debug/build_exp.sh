#!/bin/bash
# mobileposer_amd/libmp_exp<N>.so: the library with mp_lstm_persist.hip compiled with -DMP_EXP=<N> (timing experiments,
# results may be wrong by design); the other objects are the ones of the normal build.  Used through MP_LIB_PATH.
# EXTRA="-DMP_TAGX=0" SUFFIX=flags bash build_exp.sh 0  ->  libmp_exp0flags.so (the flagged hand-off of rounds 2-4)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $ROOT/mobileposer_amd/csrc/mp_lstm_persist.hip -o /tmp/mp_lstm_persist_exp$n$SUFFIX.o \
     -I $ROOT/include -Xclang -target-feature -Xclang -packed-fp32-ops -DMP_EXP=$n $EXTRA &
done
wait
for n in "$@"; do
  objs=$(ls $ROOT/mobileposer_amd/csrc/*.o | grep -v mp_lstm_persist.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/mobileposer_amd/libmp_exp$n$SUFFIX.so $objs /tmp/mp_lstm_persist_exp$n$SUFFIX.o
  echo built $ROOT/mobileposer_amd/libmp_exp$n$SUFFIX.so
done
