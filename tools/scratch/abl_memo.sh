export TK_HIP_LIB=$GRAFT_REPO_ROOT/tekken-rs_amd/libtekken_hip_ablate.so
python tools/memo_probe.py --vocab-fit heldout --batches 3 --ablate 0,65536,327680,589824,851968,1024,256,512 2>&1 | tail -18
