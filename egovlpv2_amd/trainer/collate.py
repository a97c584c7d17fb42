"""Masked-language-model collation of the EgoClip captions (reference trainer/trainer_egoclip.py:79,:122-130 calls
``transformers.DataCollatorForLanguageModeling(tokenizer, mlm=True, mlm_probability=0.15)``; transformers==4.30.0 is a
pinned dependency whose source is not under /root/reference).  This restates its published ``torch_mask_tokens``:
15 % of the non-special positions become labels; of those 80 % are replaced by <mask>, 10 % by a uniformly random token,
10 % are kept.  The draws consume the torch generator in the same order (bernoulli(p), bernoulli(0.8), bernoulli(0.5),
randint(vocab)), which tests/test_host_next_rows.py pins against the installed transformers collator."""
import torch

ROBERTA_SPECIAL_IDS = (0, 1, 2)          # <s>, <pad>, </s>
ROBERTA_MASK_ID = 50264
ROBERTA_VOCAB = 50265


def mlm_collate(input_ids: torch.Tensor, mlm_probability: float = 0.15, special_ids=ROBERTA_SPECIAL_IDS,
                mask_id: int = ROBERTA_MASK_ID, vocab_size: int = ROBERTA_VOCAB, generator=None):
    """input_ids (B, L) int64 on the host.  Returns {'input_ids': masked ids, 'labels': original id at the picked
    positions, -100 elsewhere} -- the two tensors the trainer stores as text_mlm_ids / text_mlm_labels."""
    inputs = input_ids.clone()
    labels = inputs.clone()
    prob = torch.full(labels.shape, mlm_probability)
    special = torch.zeros_like(labels, dtype=torch.bool)
    for s in special_ids:
        special |= labels == s
    prob.masked_fill_(special, value=0.0)
    masked = torch.bernoulli(prob, generator=generator).bool()
    labels[~masked] = -100
    replaced = torch.bernoulli(torch.full(labels.shape, 0.8), generator=generator).bool() & masked
    inputs[replaced] = mask_id
    rand = torch.bernoulli(torch.full(labels.shape, 0.5), generator=generator).bool() & masked & ~replaced
    words = torch.randint(vocab_size, labels.shape, dtype=torch.long, generator=generator)
    inputs[rand] = words[rand]
    return {'input_ids': inputs, 'labels': labels}
