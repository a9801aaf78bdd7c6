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
  * lamb_oracle      -- PARITY UNPINNED: the reference has no CPU implementation and no
                        numeric test of multi_tensor_lamb.cu; the oracle follows the .cu
                        line by line (citations in the file) and is cross-checked against
                        an independent closed-form LAMB step only.
"""
