# on-box experiment: in-kernel phase timers (EXP_PROF build) of the masked MFMA launches; extra flags as arguments
cd lidar-global-registration_amd/csrc
rm -f lgr_match.o; make EXP="-DEXP_PROF $*" > /dev/null 2>&1
cd ../..
LGR_MATCH_DEBUG=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep "\[lgr\] prof" | tail -2
cd lidar-global-registration_amd/csrc
rm -f lgr_match.o; make > /dev/null 2>&1
