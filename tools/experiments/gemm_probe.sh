# usage (on the GPU box): bash tools/experiments/gemm_probe.sh [phases | phases_skeleton]
# `phases`: wave 0 of workgroup 0 stamps s_memtime at the kernel's phase markers (// [probe:N] in csrc/sgemm.hip); the last
# stage's stamps survive.  Without it: whole-launch time of variants without the MFMAs / the global loads.
R=$GRAFT_REPO_ROOT
cd /tmp
if [ "$1" = phases ] || [ "$1" = phases_skeleton ]; then
  cp $R/a-link_amd/csrc/sgemm.hip sgemm_v.hip
  if [ "$1" = phases_skeleton ]; then      # the same stamps on the kernel WITHOUT its MFMAs and global loads: what the skeleton alone costs
    sed -i 's/acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av\[kk >> 1\], bv\[kk >> 1\], acc, 0, 0, 0);/acc[kk \& 15] += av[kk >> 1] * bv[kk >> 1];/' sgemm_v.hip
    sed -i 's/const f32x4 t = \*(const f32x4\*)(base + (ok ? off : 0));/const f32x4 t = f32x4{(float)(ok ? off : 0), 1.f, 2.f, 3.f};/' sgemm_v.hip
  fi
  sed -i 's|// \[probe:\([3456]\)\]|if (k0 == kbeg \&\& threadIdx.x == 0 \&\& blockIdx.x == 0 \&\& blockIdx.y == 0 \&\& blockIdx.z == 0) g_probe[\1] = __builtin_readcyclecounter();|' sgemm_v.hip
  sed -i 's|// \[probe:\([0-9]*\)\]|if (threadIdx.x == 0 \&\& blockIdx.x == 0 \&\& blockIdx.y == 0 \&\& blockIdx.z == 0) g_probe[\1] = __builtin_readcyclecounter();|' sgemm_v.hip
  sed -i 's|^namespace alink {$|__device__ long long g_probe[16];\nnamespace alink {|' sgemm_v.hip
  cat sgemm_v.hip $R/tools/experiments/gemm_probe.hip > one.hip          # one translation unit: the harness reads g_probe
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPROBE_PHASES -I$R/include -I$R/a-link_amd/csrc one.hip -o probe_ph 2>/dev/null || echo build failed
  ./probe_ph
  exit 0
fi
for V in full nomfma noloads nomfma_noloads; do
  cp $R/a-link_amd/csrc/sgemm.hip sgemm_v.hip
  case $V in
    nomfma|nomfma_noloads) sed -i 's/acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av\[kk >> 1\], bv\[kk >> 1\], acc, 0, 0, 0);/acc[kk \& 15] += av[kk >> 1] * bv[kk >> 1];/' sgemm_v.hip ;;
  esac
  case $V in
    noloads|nomfma_noloads) sed -i 's/const f32x4 t = \*(const f32x4\*)(base + (ok ? off : 0));/const f32x4 t = f32x4{(float)(ok ? off : 0), 1.f, 2.f, 3.f};/' sgemm_v.hip ;;
  esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$R/include -I$R/a-link_amd/csrc $R/tools/experiments/gemm_probe.hip sgemm_v.hip -o probe_$V 2>/dev/null || echo build failed $V
  echo "== $V"; ./probe_$V
done
