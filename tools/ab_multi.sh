cd $GRAFT_REPO_ROOT
run() { PINN_HIP_LIB=$PWD/pinn_depthestimation_amd/$1 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for rep in 1 2; do
echo "pe10x10 base $(run libpinn_hip.so '--workload pe10x10')"; echo "pe10x10 nm_w16 $(run libpinn_hip_nm_pinn_fused_w16.so '--workload pe10x10')"
echo "co100x20 base $(run libpinn_hip.so '--workload co100x20')"; echo "co100x20 nm_w32 $(run libpinn_hip_nm_pinn_fused_w32.so '--workload co100x20')"
echo "ns12x256 base $(run libpinn_hip.so '--workload ns12x256 --steps 3 --warmup 1')"; echo "ns12x256 nm_wide $(run libpinn_hip_nm_pinn_wide_w256.so '--workload ns12x256 --steps 3 --warmup 1')"
echo "ns12x256bf base $(run libpinn_hip.so '--workload ns12x256 --bf16 --steps 3 --warmup 1')"; echo "ns12x256bf nm_wide $(run libpinn_hip_nm_pinn_wide_w256.so '--workload ns12x256 --bf16 --steps 3 --warmup 1')"
done
for lib in libpinn_hip.so libpinn_hip_nm_pinn_fused_coop.so libpinn_hip.so libpinn_hip_nm_pinn_fused_coop.so; do echo "coop243 $lib $(PINN_HIP_LIB=$PWD/pinn_depthestimation_amd/$lib REPS=1000 timeout -k 10 100 python tools/bench_phases.py 243 2>&1 | grep residual_loss_grad | cut -c1-50)"; done
