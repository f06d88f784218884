"""Text helpers of the reference's ``multimnist/utils.py`` (same names and semantics)."""
import torch

max_length = 4                      # multimnist/utils.py:14
all_characters = '0123456789'
n_characters = len(all_characters)
SOS = n_characters
FILL = n_characters + 1
n_characters += 2


def char_tensor(string):
    """multimnist/utils.py:22-31: digits of ``string`` then FILL up to ``max_length`` (LongTensor)."""
    tensor = torch.full((max_length,), FILL, dtype=torch.long)
    for c, ch in enumerate(string):
        tensor[c] = all_characters.index(ch)
    return tensor


def charlist_tensor(charlist):
    """multimnist/utils.py:34-37: a list of digit indices -> padded LongTensor."""
    return char_tensor(''.join(str(int(i)) for i in charlist))


def index_to_char(top_i):
    """multimnist/utils.py:49-55"""
    top_i = int(top_i)
    if top_i == SOS:
        return '^'
    if top_i == FILL:
        return ''
    return all_characters[top_i]


def tensor_to_string(tensor):
    """multimnist/utils.py:40-46"""
    return ''.join(index_to_char(tensor[i]) for i in range(tensor.size(0)))
