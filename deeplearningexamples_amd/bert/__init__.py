"""BERT-Large pre-training (PyTorch/LanguageModeling/BERT) train-step path on MI355X."""
