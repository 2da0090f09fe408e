run() { python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_roofline --no_parity --no_alt_precision --no_entrypoint "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$*', '|', j['value'], 'img/s', j['ms_per_step'], 'ms', j.get('step_algorithmic_tflops'))"; }
run
run --cfg df_gan_damsm.yml
run --spec_norm
run --batch 64
run --imsize 128 --batch 64
run --imsize 128 --batch 64 --cfg concept_in_df_gan_damsm_nomagp.yml
run --imsize 128 --batch 64 --cfg concept_out_df_gan_sbert_damsm_nomagp.yml
run --imsize 128 --batch 64 --gen CONCEPT_OUTATTN_GEN
run --imsize 64 --batch 64
