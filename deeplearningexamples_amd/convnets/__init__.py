"""ResNet-50 v1.5 (PyTorch/Classification/ConvNets) train-step path on MI355X."""
