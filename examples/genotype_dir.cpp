// examples/genotype_dir.cpp — one locus from a Locityper-layout directory through the C ABI, files in, files out (no Python):
//
//   <root>/DB/loci/<locus>/haplotypes.fa.gz      alleles                          (lcty_fasta_read)
//   <root>/DB/loci/<locus>/kmers.bin.lz4 | .br   off-target k-mer counts          (lcty_io_read_file + lcty_kmer_counts_parse)
//   <root>/PREPROC/distr.gz                      background distributions         (lcty_io_read_file + lcty_bg_from_json)
//   <root>/OUT/loci/<locus>/aln.bam              the mapper's output              (lcty_bam_read)
//   <root>/DB/loci/<locus>/haplotypes.paf[.gz]   optional: pairwise haplotype alignments (lcty_paf_read + lcty_locus_set_hap_alns +
//                                                lcty_recover_alignments: the mapper saw the basis haplotypes only)
//        -> lcty_locus_create, lcty_reads_create / append, lcty_score_reads, lcty_solve (default scheme)
//   <root>/OUT/loci/<locus>/res.json.gz          the genotype call                (lcty_res_to_json + lcty_io_write_gz)
//   <root>/OUT/loci/<locus>/alns/00.bam (+ .bai) read placements on the call      (lcty_assignment_counts + lcty_write_bam)
//
// This is what analyze_locus (src/command/genotype.rs:1161-1260) does between the mapper and the output files, and the call sequence
// INTEGRATION.md describes for the Rust side. Build: see tests/test_gpu_example.py.   ./genotype_dir <root> <locus> [seed]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "locityper_hip.h"

static void ok(int32_t rc, const char* what) {
    if (rc != LCTY_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, lcty_last_error()); std::exit(1); }
}

static bool exists(const std::string& p) { FILE* f = std::fopen(p.c_str(), "rb"); if (f) std::fclose(f); return f != nullptr; }

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: genotype_dir <root> <locus> [seed]\n"); return 2; }
    const std::string root = argv[1], locus = argv[2];
    const uint64_t seed = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 1;
    const std::string db = root + "/DB/loci/" + locus, outd = root + "/OUT/loci/" + locus;

    // background distributions
    uint8_t* buf = nullptr; uint64_t len = 0;
    ok(lcty_io_read_file((root + "/PREPROC/distr.gz").c_str(), &buf, &len), "distr.gz");
    lcty_bg bg; double read_len = 0;
    ok(lcty_bg_from_json(reinterpret_cast<const char*>(buf), len, &bg, &read_len), "BgDistr::load");
    lcty_io_free(buf);

    // alleles
    uint32_t A = 0; uint64_t names_len = 0, seqs_len = 0;
    const std::string fa = db + "/haplotypes.fa.gz";
    ok(lcty_fasta_read(fa.c_str(), &A, nullptr, &names_len, nullptr, &seqs_len, nullptr), "haplotypes.fa.gz");
    std::vector<char> name_blob(names_len); std::vector<uint8_t> seqs(seqs_len); std::vector<uint64_t> seq_off(A + 1);
    ok(lcty_fasta_read(fa.c_str(), &A, name_blob.data(), &names_len, seqs.data(), &seqs_len, seq_off.data()), "haplotypes.fa.gz");
    std::vector<const char*> names;
    for (uint64_t i = 0, n = 0; n < A; n++) { names.push_back(&name_blob[i]); while (name_blob[i]) i++; i++; }

    // off-target k-mer counts: the first block of kmers.bin (add.rs:647-650)
    const std::string kp = exists(db + "/kmers.bin.lz4") ? db + "/kmers.bin.lz4" : db + "/kmers.bin.br";
    ok(lcty_io_read_file(kp.c_str(), &buf, &len), "kmers.bin");
    uint32_t k = 0, n_contigs = 0; uint64_t consumed = 0;
    ok(lcty_kmer_counts_parse(buf, len, &k, &n_contigs, nullptr, 0, nullptr, 0, &consumed), "KmerCounts::load (sizes)");
    if (n_contigs != A) { std::fprintf(stderr, "kmers.bin has %u contigs, the FASTA %u\n", n_contigs, A); return 1; }
    std::vector<uint64_t> cnt_off(A + 1);
    uint64_t n_counts = 0;
    for (uint32_t a = 0; a < A; a++) n_counts += seq_off[a + 1] - seq_off[a] + 1 - k;
    std::vector<uint16_t> counts(n_counts);
    ok(lcty_kmer_counts_parse(buf, len, &k, &n_contigs, cnt_off.data(), A, counts.data(), n_counts, &consumed), "KmerCounts::load");
    lcty_io_free(buf);

    // the mapper's alignments
    lcty_bam_table* bam = nullptr;
    ok(lcty_bam_read((outd + "/aln.bam").c_str(), names.data(), A, bg.is_paired, &bam), "aln.bam");
    lcty_reads_host table; const uint64_t* name_off = nullptr; const char* read_names = nullptr; uint32_t n_refs = 0;
    ok(lcty_bam_table_view(bam, &table, &name_off, &read_names, &n_refs), "bam view");

    lcty_params prm;
    lcty_params_default(&prm);
    prm.strict_subset = n_refs < A;                              // locs.rs:486
    ok(lcty_params_resolve(&prm, &bg), "params");
    lcty_ctx* ctx = nullptr;
    ok(lcty_ctx_create(0, &ctx), "context");
    lcty_locus* loc = nullptr;
    ok(lcty_locus_create(ctx, A, seqs.data(), seq_off.data(), counts.data(), cnt_off.data(), k, &bg, &prm, &loc), "locus");
    const uint64_t R = table.n_pairs;
    lcty_reads* reads = nullptr;
    ok(lcty_reads_create(loc, R, table.mate_off[2 * R], table.aln_off[R], table.cigar_off[R], &reads), "reads");
    ok(lcty_reads_append(reads, &table), "append");
    ok(lcty_score_reads(reads), "AllAlignments::load");

    // haplotypes.paf[.gz]: the mapper saw the basis haplotypes only; the alignments reach the other alleles through the pairwise
    // haplotype alignments (process_paf + HapAlns::transfer_alignments, genotype.rs:1131-1160, 149-150: --transfer 0.1 100)
    uint64_t n_recovered = 0;
    bool recovered = false;
    std::vector<uint32_t> dist;                                  // contig_distances: from the PAF ("edit") or distances.bin ("minim-div")
    bool edit_distances = false;
    for (const char* paf_name : {"/haplotypes.paf.gz", "/haplotypes.paf"}) {
        const std::string paf = db + paf_name;
        if (FILE* f = std::fopen(paf.c_str(), "rb")) std::fclose(f); else continue;
        uint64_t n_ent = 0, n_words = 0;
        ok(lcty_paf_read(paf.c_str(), names.data(), A, &n_ent, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &n_words, nullptr), "haplotypes.paf (size)");
        std::vector<uint32_t> id1(n_ent + 1), id2(n_ent + 1), nm(n_ent + 1), al(n_ent + 1), words(n_words + 1);
        std::vector<uint64_t> woff(n_ent + 1);
        dist.assign(static_cast<size_t>(A) * A, 0u);
        edit_distances = true;
        ok(lcty_paf_read(paf.c_str(), names.data(), A, &n_ent, id1.data(), id2.data(), nm.data(), al.data(), woff.data(), words.data(), &n_words, dist.data()), "haplotypes.paf");
        ok(lcty_locus_set_hap_alns(loc, static_cast<uint32_t>(n_ent), id1.data(), id2.data(), woff.data(), words.data(), nm.data(), al.data(), 100, 0.1), "HapAlns");
        ok(lcty_recover_alignments(reads, &n_recovered), "transfer_alignments");
        ok(lcty_score_reads(reads), "AllAlignments::load after recovery");
        recovered = true;
        break;
    }

    if (dist.empty()) {                                          // genotype.rs:1229-1237
        const std::string dfile = db + "/distances.bin";
        if (FILE* f = std::fopen(dfile.c_str(), "rb")) {
            std::fclose(f);
            uint8_t* dbuf = nullptr; uint64_t dlen = 0;
            ok(lcty_io_read_file(dfile.c_str(), &dbuf, &dlen), "distances.bin");
            dist.assign(static_cast<size_t>(A) * A, 0u);
            ok(lcty_distances_parse(dbuf, dlen, A, nullptr, nullptr, dist.data()), "load_divergences");
            lcty_io_free(dbuf);
        }
    }

    // solve::solve with the default scheme
    lcty_stage stages[2]; uint32_t n_stages = 0;
    ok(lcty_stages_default(stages, &n_stages), "Scheme::default");
    const uint64_t G = lcty_count_genotypes(A, 2);
    std::vector<uint16_t> gts(G * 2);
    ok(lcty_generate_genotypes(A, 2, gts.data(), G), "genotypes");
    std::vector<double> mean(G), var(G); std::vector<uint32_t> att(G);
    lcty_call call;
    ok(lcty_solve(reads, 2, stages, n_stages, seed, nullptr, &call, mean.data(), var.data(), att.data()), "solve");

    // res.json.gz
    std::vector<uint16_t> out_gts(call.n_out * 2); std::vector<double> om(call.n_out), ov(call.n_out);
    for (uint64_t t = 0; t < call.n_out; t++) {
        out_gts[2 * t] = gts[2 * call.ixs[t]]; out_gts[2 * t + 1] = gts[2 * call.ixs[t] + 1];
        om[t] = mean[call.ixs[t]]; ov[t] = var[call.ixs[t]];
    }
    // dist_to_primary / weight_dist (find_weighted_dist, solve.rs:621-645) when the contig distances are known
    std::vector<uint32_t> gdist(call.n_out + 1);
    double wdist = NAN; uint32_t warn2 = 0;
    if (!dist.empty())
        ok(lcty_call_checks(out_gts.data(), call.n_out, 2, call.ln_probs, static_cast<uint32_t>(call.n_good), dist.data(), A, gdist.data(), &wdist, &warn2), "find_weighted_dist");
    const uint32_t* gd = dist.empty() ? nullptr : gdist.data();
    uint64_t need = 0;
    ok(lcty_res_to_json(&call, out_gts.data(), 2, names.data(), A, om.data(), ov.data(), gd, edit_distances, wdist, nullptr, 0, &need), "to_json (size)");
    std::vector<char> json(need);
    ok(lcty_res_to_json(&call, out_gts.data(), 2, names.data(), A, om.data(), ov.data(), gd, edit_distances, wdist, json.data(), need, &need), "to_json");
    ok(lcty_io_write_gz((outd + "/res.json.gz").c_str(), reinterpret_cast<const uint8_t*>(json.data()), need - 1), "res.json.gz");

    // read placements on the call: the per-read posteriors of the last stage's solver, as `--out-bams 1` (solve.rs:960-973)
    const uint16_t* best = &out_gts[0];
    const uint32_t attempts = stages[n_stages - 1].attempts;
    std::vector<uint64_t> seeds(attempts);
    ok(lcty_chain_seeds(seed + 77, attempts, seeds.data()), "seeds");
    std::vector<uint64_t> read_off(call.n_good + 1);
    uint64_t n_cnt = 0;
    ok(lcty_assignment_counts(reads, best, 2, &stages[n_stages - 1].solver, attempts, seeds.data(), read_off.data(), nullptr, 0, &n_cnt), "counts (size)");
    std::vector<uint16_t> cnts(n_cnt ? n_cnt : 1);
    ok(lcty_assignment_counts(reads, best, 2, &stages[n_stages - 1].solver, attempts, seeds.data(), read_off.data(), cnts.data(), n_cnt, &n_cnt), "counts");
    uint64_t n_rec = 0;
    // after alignment recovery the batch's record table (the mapper's records + the transferred alignments) comes back from the device
    std::vector<uint64_t> m_aln_off, m_cig_off; std::vector<lcty_aln_rec> m_recs; std::vector<uint32_t> m_cigar;
    if (recovered) {
        m_aln_off.resize(R + 1); m_cig_off.resize(R + 1);
        ok(lcty_reads_get_records(reads, m_aln_off.data(), nullptr, 0, m_cig_off.data(), nullptr, 0), "records (size)");
        m_recs.resize(m_aln_off[R] + 1); m_cigar.resize(m_cig_off[R] + 1);
        ok(lcty_reads_get_records(reads, m_aln_off.data(), m_recs.data(), m_recs.size(), m_cig_off.data(), m_cigar.data(), m_cigar.size()), "records");
        table.aln_off = m_aln_off.data(); table.recs = m_recs.data(); table.cigar_off = m_cig_off.data(); table.cigar = m_cigar.data();
    }
    ok(lcty_write_bam((outd + "/alns/00.bam").c_str(), reads, &table, name_off, read_names, nullptr, nullptr, names.data(), best, 2,
                      static_cast<uint16_t>(attempts), read_off.data(), cnts.data(), &n_rec), "write_bam");

    std::printf("genotype %s,%s quality %.1f reads %llu unexplained %u warnings %u bam_records %llu recovered %llu\n", names[best[0]], names[best[1]],
                call.quality, static_cast<unsigned long long>(call.n_good), call.unexpl_reads, call.warnings, static_cast<unsigned long long>(n_rec),
                static_cast<unsigned long long>(n_recovered));
    lcty_reads_destroy(reads); lcty_bam_table_free(bam); lcty_locus_destroy(loc); lcty_ctx_destroy(ctx);
    return 0;
}
