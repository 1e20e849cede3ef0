#!/bin/bash
# full profile refresh of the round: kernel trace + HBM PMC passes of the step, MFMA / VALU utilisation, K1 encode (zipf, uniform)
cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
bash tools/profile_mfma.sh > gpurun_out/mfma.log 2>&1
bash tools/profile_encode.sh zipf > gpurun_out/encode_zipf.log 2>&1
bash tools/profile_encode.sh uniform > gpurun_out/encode_uniform.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/make_profiles.py $TAG
python tools/make_mfma_profile.py $TAG
python tools/make_encode_profile.py $TAG zipf
python tools/make_encode_profile.py $TAG uniform
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
rm -rf gpurun_out/refresh/ktrace gpurun_out/refresh/recipe gpurun_out/refresh/fetch gpurun_out/refresh/write gpurun_out/mfma/mfma gpurun_out/mfma/valu gpurun_out/encode_*/ktrace gpurun_out/encode_*/fetch gpurun_out/encode_*/write
python bench.py > gpurun_out/profiles_$TAG/${TAG}_bench_line.json 2> gpurun_out/bench_final.err
tail -c 300 gpurun_out/bench_final.err
ls -la gpurun_out/profiles_$TAG
