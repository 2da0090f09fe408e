# the f16 MA-GP iteration test under kernel choices that are bit-different but equally exact (run on a GPU box from the repo root):
# errG_fake against the f32 oracle moved between 1.8e-4 and 1.9e-3 in round 4 (the generator-step losses follow the PENALTY's Adam step)
T='tests/test_precision_modes_gpu.py::test_f16_mode_losses_within_1e_3_of_f32_reference[df_gan_damsm.yml-synth]'
for sw in "" "no_ptile_slab128" "no_dstem" "no_stage_bits" "no_ptile_slab128,no_dstem"; do
  XMC_DEBUG_DISPATCH=$sw timeout -k 10 200 python -m pytest "$T" -x -q -s 2>&1 | grep -E "errG_fake: |losses vs f32|passed|failed" | cut -c1-220 | sed "s/^/[$sw] /"
done
