// tests/host_blocks_main.cpp -- drives the C++ host mirror (gr-dvbs2rx_amd/host/dvbs2rx_hip_blocks.h) the way a
// GNU Radio scheduler thread would: forecast() + general_work() on byte streams read from files written by
// tests/test_host_blocks.py, which then compares the output streams with the CPU oracle.
//   usage: host_blocks_main <ldpc|bch|bbdh|demap|loop|chain> <in file> <out file> <framesize> <rate name> <arg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include "../gr-dvbs2rx_amd/host/dvbs2rx_hip_blocks.h"
using namespace dvbs2rx_hip;

static std::vector<char> slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::vector<char>((std::istreambuf_iterator<char>(f)), {}); }

int main(int argc, char** argv)
{
    if (argc < 7) return 2;
    const std::string kind = argv[1];
    std::vector<char> in = slurp(argv[2]);
    const dvb_framesize_t fs = (dvb_framesize_t)atoi(argv[4]);
    const int rate = dvbs2_rate_from_name(argv[5]);
    const int arg = atoi(argv[6]);
    std::vector<char> out;
    int consumed = 0, produced = 0;
    gr_vector_int ninput(1), req(1);
    gr_vector_const_void_star ii(1, in.data());
    gr_vector_void_star oo(1);
    try {
        if (kind == "ldpc") {
            auto b = ldpc_decoder_bb::make(STANDARD_DVBS2, fs, rate, MOD_QPSK, OM_MESSAGE, INFO_OFF, arg, 0, 32, 64);
            int nout = b->output_multiple() * 2; // two reference batches
            b->forecast(nout, req);
            if ((size_t)req[0] > in.size()) { std::fprintf(stderr, "short input %d > %zu\n", req[0], in.size()); return 3; }
            out.resize(nout); oo[0] = out.data();
            uint64_t pdu_frames = 0;
            b->set_llr_pdu_handler([&](uint64_t fc, int simd, const int8_t*, size_t n) { pdu_frames += simd; (void)fc; (void)n; });
            produced = b->general_work(nout, ninput, ii, oo, &consumed);
            std::printf("avg_trials %u pdu_frames %llu\n", b->get_average_trials(), (unsigned long long)pdu_frames);
        } else if (kind == "bch") {
            auto b = bch_decoder_bb::make(STANDARD_DVBS2, fs, rate, OM_MESSAGE);
            int nout = b->output_multiple() * arg;
            b->forecast(nout, req);
            out.resize(nout); oo[0] = out.data();
            produced = b->general_work(nout, ninput, ii, oo, &consumed);
            std::printf("frames %llu errors %llu\n", (unsigned long long)b->get_frame_count(), (unsigned long long)b->get_error_count());
        } else if (kind == "bbdh") {
            // bbdeheader_bb: `arg` BBFRAMEs in two general_work calls (the block's state must carry over)
            auto b = bbdeheader_bb::make(STANDARD_DVBS2, fs, rate);
            const int fb = (int)(in.size() / arg), first = arg / 2;
            out.resize((size_t)arg * b->output_multiple() + 188);
            int c = 0;
            ninput[0] = first * fb; oo[0] = out.data();
            produced = b->general_work(first * b->output_multiple(), ninput, ii, oo, &c);
            consumed = c;
            ninput[0] = (arg - first) * fb; ii[0] = in.data() + consumed; oo[0] = out.data() + produced;
            produced += b->general_work((arg - first) * b->output_multiple(), ninput, ii, oo, &c);
            consumed += c;
            std::printf("packets %llu errors %llu bbframes %llu dropped %llu gaps %llu\n", (unsigned long long)b->get_packet_count(),
                        (unsigned long long)b->get_error_count(), (unsigned long long)b->get_bbframe_count(),
                        (unsigned long long)b->get_bbframe_drop_count(), (unsigned long long)b->get_bbframe_gap_count());
        } else if (kind == "loop") {
            // demapper -> LDPC with the llr_pdu port wired back into the demapper (apps/dvbs2-rx:853-863, 873):
            // 32 frames demapped with the pre-decoder estimate, decoded, refined; then 32 more with the refined N0.
            auto dm = xfecframe_demapper_cb::make(fs, rate, (dvb_constellation_t)arg, 64);
            auto ld = ldpc_decoder_bb::make(STANDARD_DVBS2, fs, rate, (dvb_constellation_t)arg, OM_CODEWORD, INFO_OFF, 25, 0, 32, 64);
            int found = 0;
            ld->set_llr_pdu_handler([&](uint64_t fc, int simd, const int8_t* llr, size_t n) { found += dm->handle_llr_pdu(fc, simd, llr, n); });
            const int nllr = dm->output_multiple() * 32;
            std::vector<char> llr(nllr);
            std::vector<char> bits(ld->output_multiple());
            out.resize(2 * nllr);
            for (int round = 0; round < 2; round++) {
                dm->forecast(nllr, req);
                gr_vector_const_void_star di(1, in.data() + (size_t)round * req[0] * 8);
                gr_vector_void_star dout(1, out.data() + (size_t)round * nllr);
                int c1 = 0, c2 = 0;
                produced += dm->general_work(nllr, ninput, di, dout, &c1);
                consumed += c1;
                if (round == 0) {
                    gr_vector_const_void_star li(1, dout[0]);
                    gr_vector_void_star lo(1, bits.data());
                    ld->general_work((int)bits.size(), ninput, li, lo, &c2);
                    std::printf("found %d snr_lin %.6f\n", found, std::pow(10.0, dm->get_snr() / 10.0));
                }
            }
        } else if (kind == "chain") {
            // the three blocks one after the other on HOST buffers (what GNU Radio's scheduler does, apps/dvbs2-rx:853-863) against ONE call
            // of the fused host-pointer entry dvbs2_chain_decode on the same symbols; `arg` = constellation, fixed N0 = 1 / 7.0
            const dvb_constellation_t mod = (dvb_constellation_t)arg;
            auto dm = xfecframe_demapper_cb::make(fs, rate, mod, 256);
            auto ld = ldpc_decoder_bb::make(STANDARD_DVBS2, fs, rate, mod, OM_MESSAGE, INFO_OFF, 25, 0, 32, 256);
            auto bc = bch_decoder_bb::make(STANDARD_DVBS2, fs, rate, OM_MESSAGE);
            dm->set_snr_lin(7.0f);
            gr_vector_int r1(1);
            dm->forecast(dm->output_multiple(), r1);
            const int n_frames = (int)(in.size() / ((size_t)r1[0] * 8));
            std::vector<char> llr((size_t)n_frames * dm->output_multiple()), bits((size_t)n_frames * (ld->output_multiple() / 32)); // (the LDPC block's granule is one group of 32 frames: n_frames must be a multiple of it)
            out.resize((size_t)n_frames * bc->output_multiple());
            int c = 0;
            { gr_vector_void_star o1(1, llr.data()); dm->general_work((int)llr.size(), ninput, ii, o1, &c); consumed = c; }
            { gr_vector_const_void_star i2(1, llr.data()); gr_vector_void_star o2(1, bits.data()); ld->general_work((int)bits.size(), ninput, i2, o2, &c); }
            { gr_vector_const_void_star i3(1, bits.data()); oo[0] = out.data(); produced = bc->general_work((int)out.size(), ninput, i3, oo, &c); }
            dvbs2_chain_t* ch = nullptr;
            if (dvbs2_chain_create(&ch, STANDARD_DVBS2, fs, rate, mod, 32, n_frames, 0) != DVBS2_OK) { std::printf("chain create: %s\n", dvbs2_last_error()); return 5; }
            std::vector<unsigned char> fused(out.size());
            std::vector<int32_t> ret((n_frames + 31) / 32), corr(n_frames);
            const float n0 = 1.0f / 7.0f;
            if (dvbs2_chain_decode(ch, reinterpret_cast<const float*>(in.data()), n_frames, &n0, 1, 25, fused.data(), ret.data(), corr.data()) != DVBS2_OK) { std::printf("chain decode: %s\n", dvbs2_last_error()); return 5; }
            dvbs2_chain_destroy(ch);
            int bad = 0;
            for (int f = 0; f < n_frames; f++) bad += corr[f] < 0;
            std::printf("frames %d fused_equals_blocks %d bch_failures %d block_errors %llu\n", n_frames, (int)(std::memcmp(fused.data(), out.data(), out.size()) == 0), bad,
                        (unsigned long long)bc->get_error_count());
        } else {
            auto b = xfecframe_demapper_cb::make(fs, rate, (dvb_constellation_t)arg);
            int nout = b->output_multiple() * 2;
            b->forecast(nout, req);
            out.resize(nout); oo[0] = out.data();
            b->set_snr_lin(10.0f);
            produced = b->general_work(nout, ninput, ii, oo, &consumed);
            std::printf("snr_db %.3f\n", b->get_snr());
        }
    } catch (const std::exception& e) { std::printf("exception: %s\n", e.what()); return 4; }
    std::printf("consumed %d produced %d\n", consumed, produced);
    std::ofstream(argv[3], std::ios::binary).write(out.data(), produced);
    return 0;
}
