cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc1 -- python tools/bench_kernels.py --iters 2 --only nt,tn > gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d gpurun_out/pmc2 -- python tools/bench_kernels.py --iters 2 --only nt,tn > gpurun_out/pmc2.log 2>&1
ls gpurun_out/pmc1/* gpurun_out/pmc2/* | head; tail -3 gpurun_out/pmc2.log
