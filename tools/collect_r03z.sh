# copy the summaries of gpurun_out/r03z (tools/gpu_r03z.sh) into profiles/ under their r03z_ names
O=gpurun_out/r03z
for f in bench.json bench_direct_kernels_only.json bench_fcos_b3_832x1344.json bench_fcos_r50.json bench_gn_unfused.json bench_latency_b1_512.json bench_latency_b1_512_graph.json bench_latency_b2_512.json bench_latency_b2_512_graph.json bench_mnfcos.json bench_train.json bench_train_amp.json layer_times.tsv layer_times_fcos_b3.tsv rccl_world1.json rehearsal_gloo2_infer.json rehearsal_gloo2_train.json rocprofv3_kernel_stats.csv smoke.log train_step_amp_kernel_stats.csv train_step_kernel_stats.csv pmc_summary.json; do cp $O/$f profiles/r03z_$f; done
cp $O/pytest.log profiles/r03z_pytest_gpu.log; cp $O/pmc_traffic.json profiles/pmc_traffic.json
