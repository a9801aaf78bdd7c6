"""CPU oracle for the hot path (TEST INFRASTRUCTURE ONLY).

Everything under ``oracle/`` is a CPU restatement of the reference's algorithm for
the AMP+DDP train-step hot path.  It exists so that the HIP
kernels can be checked; it is never the thing measured or shipped.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product package ``deeplearningexamples_amd`` must not.

Pinning status (see DESIGN.md "Oracle"):
  * dlrm_oracle      -- pinned: bit-exact against the reference's own Python
                        (dlrm/utils/distributed.py, dlrm/nn/interactions.py,
                        dlrm/nn/embeddings.py) imported in the build container,
                        vectors committed under tests/golden/ by oracle/make_golden.py.
  * bert_oracle      -- pinned against the reference's eager modeling.py
                        (BertForPreTraining, tiny config + loss fixtures).
  * resnet_oracle    -- pinned against the reference's image_classification.models.resnet50
                        eager module (loss / grad-norm fixtures).
  * lamb_oracle      -- pinned (round 2): the reference's UNMODIFIED FusedLAMBAMP + PolyWarmUpScheduler + torch
                        GradScaler step on CPU with `fused_lamb_CUDA` bound to this module's numpy kernels
                        (oracle/lamb_cpu_ext.py) wrote tests/golden/lamb_ref_steps.npz; the oracle's host-sequence
                        restatement reproduces it bit for bit.  The per-element arithmetic of the .cu kernels has no
                        CPU counterpart in the reference; it follows multi_tensor_lamb.cu line by line and is
                        cross-checked against an independent float64 closed form.
  * philox_oracle    -- pinned by the Random123 known-answer vectors.
  * waveglow_oracle  -- section 8 row f1 (product: deeplearningexamples_amd/waveglow): the WaveGlow training loss, pinned
                        against the reference's own WaveGlow + WaveGlowLoss on CPU (loss and every parameter gradient,
                        tests/golden/waveglow_loss.npz).
  * tacotron2_oracle -- section 8 row f1 (product: deeplearningexamples_amd/tacotron2): the Tacotron2 training loss with
                        externally supplied dropout masks, pinned against the reference's own Tacotron2 + Tacotron2Loss in
                        training mode under the same masks (loss, alignments, all 60 parameter gradients,
                        tests/golden/tacotron2_loss.npz).
  * storage          -- not an oracle of the reference: 16-bit storage emulation (round-to-dtype with a
                        straight-through gradient) used by the step oracles to MEASURE the precision floor that the
                        loss-parity bars add to north_star's 1e-3.
"""
