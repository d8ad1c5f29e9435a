"""An offline HF fast tokenizer (word level: "w5", "w6", ...) standing in for the hub tokenizers the reference loads by name."""
from tokenizers import Tokenizer, models, pre_tokenizers
from transformers import PreTrainedTokenizerFast

SPECIALS = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[BOS]"]


def toy_tokenizer(vocab_size: int, truncation_side: str = "right"):
    vocab = {s: i for i, s in enumerate(SPECIALS)}
    for i in range(len(SPECIALS), vocab_size):
        vocab[f"w{i}"] = i
    t = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    t.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    return PreTrainedTokenizerFast(tokenizer_object=t, pad_token="[PAD]", unk_token="[UNK]", cls_token="[CLS]", sep_token="[SEP]",
                                   bos_token="[BOS]", eos_token="[SEP]", truncation_side=truncation_side)
