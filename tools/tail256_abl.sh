# tuning builds of csrc/conv_fuse256.hip with pieces compiled out (MHE_T256_ABL bits: 1 no MFMA, 2 no transform, 4 no weight stages, 8 no identity
# loads, 16 no a_out stores, 32 no output stores) as stand-alone libraries under mhentropy_amd/csrc/_abl/ (git-ignored; they travel with gpurun),
# timed at config C2's layer3 shape:   bash tools/tail256_abl.sh "0 1 2 4 8 16 32 63" [build]
mkdir -p mhentropy_amd/csrc/_abl
for a in $1; do
  if [ "$2" = "build" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Imhentropy_amd/csrc -DMHE_T256_ABL=$a \
       mhentropy_amd/csrc/conv_fuse256.hip mhentropy_amd/csrc/api.hip -o mhentropy_amd/csrc/_abl/libt256_abl$a.so || exit 1
  else
    echo "ABL $a: $(MHE_T256_LIB=mhentropy_amd/csrc/_abl/libt256_abl$a.so timeout -k 10 100 python tools/tail256_time.py 2>&1 | tail -1)"
  fi
done
