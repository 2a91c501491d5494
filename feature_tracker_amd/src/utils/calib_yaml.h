// Reader for the subset of OpenCV's FileStorage YAML that the stereo calibration file uses
// (build_pc/config/test.yaml, read by load_params at src/test_tracking.cpp:27-57): top-level `NAME: number`
// entries and `NAME: !!opencv-matrix` blocks with rows / cols / dt / data: [ ... ] (the list may span lines).
#pragma once
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace feature_tracker {

struct YamlMatrix {
    int rows = 0, cols = 0;
    std::vector<double> data;
    bool empty() const { return data.empty() || rows * cols != (int)data.size(); }
    double at(int r, int c) const { return data[(size_t)r * cols + c]; }
    double at(int i) const { return data[(size_t)i]; }
};

class CalibYaml {
public:
    bool open(const std::string& path) {
        std::ifstream f(path);
        if (!f.is_open()) return false;
        std::stringstream ss;
        ss << f.rdbuf();
        parse(ss.str());
        opened_ = true;
        return true;
    }
    bool isOpened() const { return opened_; }
    // missing keys read as 0 / an empty matrix, as cv::FileNode does
    double number(const std::string& key) const {
        auto it = scalars_.find(key);
        return it == scalars_.end() ? 0.0 : it->second;
    }
    YamlMatrix matrix(const std::string& key) const {
        auto it = matrices_.find(key);
        return it == matrices_.end() ? YamlMatrix() : it->second;
    }

private:
    static std::string trim(const std::string& s) {
        size_t a = 0, b = s.size();
        while (a < b && std::isspace((unsigned char)s[a])) a++;
        while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
        return s.substr(a, b - a);
    }
    static std::string strip_comment(const std::string& line) {
        const size_t h = line.find('#');
        return h == std::string::npos ? line : line.substr(0, h);
    }

    void parse(const std::string& text) {
        std::vector<std::string> lines;
        {
            std::stringstream ss(text);
            std::string l;
            while (std::getline(ss, l)) lines.push_back(l);
        }
        for (size_t i = 0; i < lines.size(); i++) {
            if (lines[i].rfind("%YAML", 0) == 0 || lines[i].rfind("---", 0) == 0) continue;
            const std::string line = strip_comment(lines[i]);
            if (line.empty() || std::isspace((unsigned char)line[0])) continue;   // only top-level keys start a node
            const size_t colon = line.find(':');
            if (colon == std::string::npos) continue;
            const std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
            if (val.find("!!opencv-matrix") != std::string::npos) {
                // the block is every following indented (or blank) line
                std::string block;
                size_t j = i + 1;
                for (; j < lines.size(); j++) {
                    const std::string b = strip_comment(lines[j]);
                    if (!b.empty() && !std::isspace((unsigned char)b[0])) break;
                    block += b;
                    block += '\n';
                }
                i = j - 1;
                YamlMatrix m;
                m.rows = (int)field(block, "rows");
                m.cols = (int)field(block, "cols");
                const size_t d = block.find("data");
                const size_t lb = d == std::string::npos ? std::string::npos : block.find('[', d);
                const size_t rb = lb == std::string::npos ? std::string::npos : block.find(']', lb);
                if (rb != std::string::npos) {
                    std::string body = block.substr(lb + 1, rb - lb - 1);
                    for (char& c : body)
                        if (c == ',') c = ' ';
                    std::stringstream ns(body);
                    double v;
                    while (ns >> v) m.data.push_back(v);
                }
                matrices_[key] = m;
            } else if (!val.empty()) {
                char* end = nullptr;
                const double v = std::strtod(val.c_str(), &end);
                if (end != val.c_str()) scalars_[key] = v;
            }
        }
    }
    static double field(const std::string& block, const char* name) {
        const size_t p = block.find(std::string(name) + ":");
        return p == std::string::npos ? 0.0 : std::strtod(block.c_str() + p + std::string(name).size() + 1, nullptr);
    }

    bool opened_ = false;
    std::map<std::string, double> scalars_;
    std::map<std::string, YamlMatrix> matrices_;
};

}  // namespace feature_tracker
