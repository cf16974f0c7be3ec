echo "== independent solves, gated"; DBG_INDEP=1 timeout -k 10 400 python tools/dbg_gate_p2p.py 6 2e6 5 2>&1 | grep -v amdgpu.ids | grep "^rep\|rror"
echo "== independent solves, BZ_GATE=0"; BZ_GATE=0 DBG_INDEP=1 timeout -k 10 400 python tools/dbg_gate_p2p.py 3 2e6 5 2>&1 | grep -v amdgpu.ids | grep "^rep\|rror"
