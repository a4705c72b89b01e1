# tuning builds of csrc/conv_halo.hip with pieces compiled out (MHE_HALO_ABL bits: 1 no MFMA, 2 no fragment reads, 4 no weight DMA,
# 8 no halo staging, 16 no output walk) as stand-alone libraries under mhentropy_amd/csrc/_abl/ (git-ignored; they travel with gpurun):
#   bash tools/halo_abl_build.sh "0 2 6 10 18 30"
mkdir -p mhentropy_amd/csrc/_abl
# a variant may carry other -D switches after commas: "0,-DNAME=1" (the library is named after the whole string)
for v in $1; do
  a=${v%%,*}; rest=${v#$a}; rest=${rest//,/ }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Imhentropy_amd/csrc -DMHE_HALO_ABL=$a $rest \
     mhentropy_amd/csrc/conv_halo.hip mhentropy_amd/csrc/api.hip -o mhentropy_amd/csrc/_abl/libhalo_abl$v.so || exit 1
done
ls -la mhentropy_amd/csrc/_abl
