set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests/test_hip_step.py -x -q -m gpu -k "pipelined" > gpurun_out/r05/t_pipe.log 2>&1 || (tail -30 gpurun_out/r05/t_pipe.log; exit 1)
tail -2 gpurun_out/r05/t_pipe.log
bash tools/ab_flags.sh -r 2 "--dtype f32" "--dtype f32 --pipeline" "--dtype f32 --pipeline --tune f32_pipe_solo=" "--dtype f32 --pipeline --tune f32_pipe_solo=enc0" "--dtype f32 --pipeline --tune f32_pipe_solo=enc0,enc1,dec1" > gpurun_out/r05/ab_f32_pipeline_solo.log 2>&1
cat gpurun_out/r05/ab_f32_pipeline_solo.log
bash tools/ab_flags.sh -r 1 "--config c5dims --dtype f32 --steps 60" "--config c5dims --dtype f32 --steps 60 --pipeline" > gpurun_out/r05/ab_f32_pipeline_solo_c5.log 2>&1
cat gpurun_out/r05/ab_f32_pipeline_solo_c5.log
