for a in $1; do echo "ABL $a"; MHE_HALO_LIB=mhentropy_amd/csrc/_abl/libhalo_abl$a.so timeout -k 10 100 python tools/halo_check.py 256 2>&1 | grep "pass " ; done
