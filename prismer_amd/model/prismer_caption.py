"""Caption head (reference: model/prismer_caption.py:14-121): training loss, answer ranking, generation."""
import torch

from .prismer import Prismer


def _first_token_probs(logits_last, first_ids):
    """softmax(logits, dim=1).index_select(1, first token of every candidate answer) (prismer_caption.py:70, prismer_vqa.py:51) through the
    library's kernel (ph_softmax_gather_bf16: no stock torch softmax on the rank path, round 6); logits_last: [B, V] view of the decoder's logits"""
    from .. import ops
    x = logits_last
    if x.dtype != torch.bfloat16 or x.stride(-1) != 1 or (x.stride(0) % 8) != 0 or (x.data_ptr() % 16) != 0:
        x = x.to(torch.bfloat16).contiguous()
        if x.shape[1] % 8:                                      # (row stride must be a multiple of 8 elements)
            x = torch.nn.functional.pad(x, (0, 8 - x.shape[1] % 8))[:, :logits_last.shape[1]]
    return ops.softmax_gather(x, first_ids.to(torch.int64))


def tile(x, dim, n_tile):
    """prismer_caption.py:115-121: repeat each slice n_tile times consecutively."""
    return x.repeat_interleave(n_tile, dim=dim)


class PrismerCaption(Prismer):
    prompt_length = 4          # len(tokenizer('A picture of').input_ids) - 1 (prismer_caption.py:24-25) when ids are passed directly

    def _ids(self, text, device, **kw):
        """strings -> tokenizer (reference behaviour); dict/tuple of tensors -> used as given (offline path)."""
        if isinstance(text, dict):
            return text['input_ids'].to(device), text['attention_mask'].to(device)
        if isinstance(text, (tuple, list)) and len(text) == 2 and torch.is_tensor(text[0]):
            return text[0].to(device), text[1].to(device)
        t = self._tokenize(text, return_tensors='pt', **kw).to(device)
        return t.input_ids, t.attention_mask

    def forward(self, experts, caption=None, answer=None, train=True, prefix='', inference='generate', k_test=32, return_scores=False):
        device = experts['rgb'].device
        pad = self.text_decoder.config.pad_token_id
        if train:
            experts_train = self.expert_encoder(experts).permute(1, 0, 2)              # 'l b d -> b l d'
            input_ids, attention_mask = self._ids(caption, device, padding='longest', truncation=True, max_length=30)
            answer_targets = input_ids.masked_fill(input_ids == pad, -100)
            if isinstance(prefix, int):
                answer_targets[:, :prefix] = -100
            elif len(prefix) > 0:
                plen = len(self._tokenize(prefix).input_ids) - 1 if self.tokenizer is not None else self.prompt_length
                answer_targets[:, :plen] = -100
            out = self.text_decoder(input_ids, attention_mask=attention_mask, encoder_hidden_states=experts_train, labels=answer_targets,
                                    return_dict=True)
            return out.loss.mean()
        with torch.no_grad():
            experts_train = self.expert_encoder(experts).permute(1, 0, 2)
            B = experts['rgb'].size(0)
            if inference == 'generate':
                if isinstance(prefix, str):
                    ids, att = self._ids([prefix] * B, device, padding='longest')
                else:
                    ids, att = self._ids(prefix, device)
                ids, att = ids[:, :-1], att[:, :-1]                                        # drop </s>
                outputs = self.text_decoder.generate(input_ids=ids, encoder_hidden_states=experts_train, attention_mask=att, num_beams=3,
                                                     max_length=20, min_length=8)       # prismer_caption.py:45-50
                if self.tokenizer is None:
                    return outputs
                captions = []
                for o in outputs:
                    cap = self.tokenizer.decode(o, skip_special_tokens=True)
                    captions.append(cap[len(prefix) + (1 if len(prefix) > 0 else 0):])
                return captions
            # inference == 'rank' (prismer_caption.py:59-112)
            if isinstance(answer, (list, tuple)) and answer and isinstance(answer[0], str):
                answer = [' ' + a.lower() + '</s>' for a in answer]
                a_ids, a_att = self._ids(answer, device, padding='longest', add_special_tokens=False)
            else:
                a_ids, a_att = self._ids(answer, device)
            if isinstance(prefix, str):
                s_ids, s_att = self._ids([prefix] * B, device, padding='longest')
            else:
                s_ids, s_att = self._ids(prefix, device)
            s_ids, s_att = s_ids[:, :-1], s_att[:, :-1]
            start = self.text_decoder(s_ids, attention_mask=s_att, encoder_hidden_states=experts_train, return_dict=True)
            prob_first = _first_token_probs(start.logits[:, -1, :], a_ids[:, 0])
            _, topk_ids = prob_first.topk(k_test, dim=1)
            ans_ids = torch.cat([a_ids.index_select(0, t) for t in topk_ids], dim=0)
            ans_att = torch.cat([a_att.index_select(0, t) for t in topk_ids], dim=0)
            input_ids = torch.cat([tile(s_ids, 0, k_test), ans_ids], dim=1).long()
            att = torch.cat([tile(s_att, 0, k_test), ans_att], dim=1)
            enc = tile(experts_train, 0, k_test)
            targets = input_ids.masked_fill(input_ids == pad, -100)
            targets[:, :-a_ids.shape[1]] = -100
            out = self.text_decoder(input_ids, attention_mask=att, encoder_hidden_states=enc, labels=targets, return_dict=True)
            lp = (-out.loss / torch.sum(targets != -100, dim=-1)).view(-1, k_test)
            best = lp.argmax(dim=1)
            if return_scores:                                   # (extension for the parity tests: the candidates and their scores)
                return topk_ids[best >= 0, best], topk_ids, lp
            return topk_ids[best >= 0, best]
