for r in 1 2; do for v in "" 1; do
R2L_NO_DW_SLAB=$v python bench.py --no-cpu-baseline --no-teacher --steps 10 2>gpurun_out/ab_err.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('noslab=$v', 'train %.3f ms' % d['fast_mode']['train']['ms_per_step'], 'train4096 %.3f ms' % d['fast_mode']['train_4096']['ms_per_step'])"
done; done

