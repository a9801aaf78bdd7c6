"""MI355X-native kernels + host harness for the NVIDIA/DeepLearningExamples AMP+DDP train-step
hot path (RN50 v1.5, BERT-Large phase-1, DLRM).  See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"
