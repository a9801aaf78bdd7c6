"""Synthetic stand-in for lddl.torch.get_bert_pretrain_data_loader: batches with the 5 int64 keys of
run_pretraining.py:603-609 (input_ids, token_type_ids, attention_mask, labels, next_sentence_labels)."""
import torch


class _SyntheticLoader:
    def __init__(self, batch_size, seq_len, vocab_size, masked_per_seq, steps, seed):
        self.b, self.s, self.v, self.k, self.n = batch_size, seq_len, vocab_size, masked_per_seq, steps
        self.g = torch.Generator().manual_seed(seed)

    def __len__(self):
        return self.n

    def __iter__(self):
        for _ in range(self.n):
            b, s = self.b, self.s
            ids = torch.randint(0, self.v, (b, s), generator=self.g)
            split = torch.randint(s // 4, 3 * s // 4, (b, 1), generator=self.g)
            labels = torch.full((b, s), -1, dtype=torch.long)
            for i in range(b):
                pos = torch.randperm(s, generator=self.g)[:self.k]
                labels[i, pos] = torch.randint(0, self.v, (self.k,), generator=self.g)
            yield {"input_ids": ids, "token_type_ids": (torch.arange(s)[None, :] >= split).long(),
                   "attention_mask": torch.ones((b, s), dtype=torch.long), "labels": labels,
                   "next_sentence_labels": torch.randint(0, 2, (b,), generator=self.g)}


def get_bert_pretrain_data_loader(path, local_rank=0, shuffle_buffer_size=16384, shuffle_buffer_warmup_factor=16,
                                  vocab_file=None, data_loader_kwargs=None, mlm_probability=0.15, base_seed=12345,
                                  log_dir=None, log_level=None, start_epoch=0, return_raw_samples=False,
                                  sequence_length_alignment=8, ignore_index=-1, **kw):
    kw2 = data_loader_kwargs or {}
    return _SyntheticLoader(kw2.get("batch_size", 8), 128, 30522, 20, kw.get("steps", 1000), base_seed + local_rank)
