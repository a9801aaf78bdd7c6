"""apex surface of the DLRM / BERT train steps: multi_tensor_apply, mlp, optimizers."""
