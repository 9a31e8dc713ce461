// ORACLE (test infrastructure, NOT product code).
// Restates the reference's ParameterServer (scpp_core/utils/include/parameterServer.hpp:34-127)
// on a minimal Boost-INFO reader: `key value`, `key { children }`, `;` comments.
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace oracle
{

struct InfoNode
{
    std::string value;
    std::vector<std::pair<std::string, std::shared_ptr<InfoNode>>> children;

    const InfoNode *child(const std::string &key) const
    {
        for (auto &c : children)
            if (c.first == key)
                return c.second.get();
        return nullptr;
    }
    size_t count(const std::string &key) const
    {
        size_t n = 0;
        for (auto &c : children)
            n += (c.first == key);
        return n;
    }
};

class ParameterServer
{
  public:
    explicit ParameterServer(const std::string &filename)
    {
        std::ifstream f(filename);
        if (!f)
            throw std::runtime_error("Could not open file for reading: " + filename);
        struct Tok
        {
            std::string s;
            int line;
        };
        std::vector<Tok> toks;
        std::string line;
        int ln = 0;
        while (std::getline(f, line))
        {
            ln++;
            const size_t sc = line.find(';');
            if (sc != std::string::npos)
                line = line.substr(0, sc);
            std::string cur;
            auto flush = [&]() {
                if (!cur.empty())
                {
                    toks.push_back({cur, ln});
                    cur.clear();
                }
            };
            for (char ch : line)
            {
                if (ch == '{' || ch == '}')
                {
                    flush();
                    toks.push_back({std::string(1, ch), ln});
                }
                else if (ch == ' ' || ch == '\t' || ch == '\r')
                    flush();
                else
                    cur.push_back(ch);
            }
            flush();
        }
        size_t pos = 0;
        std::vector<InfoNode *> stack{&root};
        while (pos < toks.size())
        {
            const Tok &t = toks[pos];
            if (t.s == "}")
            {
                if (stack.size() < 2)
                    throw std::runtime_error("INFO parse error: unmatched } in " + filename);
                stack.pop_back();
                pos++;
                continue;
            }
            if (t.s == "{")
                throw std::runtime_error("INFO parse error: unexpected { in " + filename);
            auto node = std::make_shared<InfoNode>();
            stack.back()->children.push_back({t.s, node});
            pos++;
            if (pos < toks.size() && toks[pos].line == t.line && toks[pos].s != "{" && toks[pos].s != "}")
            {
                node->value = toks[pos].s;
                pos++;
            }
            if (pos < toks.size() && toks[pos].s == "{")
            {
                stack.push_back(node.get());
                pos++;
            }
        }
    }

    // parameterServer.hpp:64-77
    void loadScalar(const std::string &name, double &out) const
    {
        const InfoNode *n = root.child(name);
        if (!n || n->value.empty())
            throw std::runtime_error("WARNING: Failed to load scalar type: " + name + "!");
        char *end = nullptr;
        out = std::strtod(n->value.c_str(), &end);
        if (end == n->value.c_str())
            throw std::runtime_error("WARNING: Failed to load scalar type: " + name + "!");
    }
    void loadScalar(const std::string &name, size_t &out) const
    {
        double d;
        loadScalar(name, d);
        out = size_t(d);
    }
    void loadScalar(const std::string &name, bool &out) const
    {
        const InfoNode *n = root.child(name);
        if (!n)
            throw std::runtime_error("WARNING: Failed to load scalar type: " + name + "!");
        if (n->value == "true" || n->value == "1")
            out = true;
        else if (n->value == "false" || n->value == "0")
            out = false;
        else
            throw std::runtime_error("WARNING: Failed to load scalar type: " + name + "!");
    }

    // parameterServer.hpp:79-127 (vectors only: every shipped config uses `(i)` keys)
    void loadVector(const std::string &name, double *out, size_t rows) const
    {
        const InfoNode *n = root.child(name);
        if (!n)
            throw std::runtime_error("Failed to load matrix type: " + name + "!");
        double scaling = 1.;
        if (const InfoNode *s = n->child("scaling"))
            scaling = std::strtod(s->value.c_str(), nullptr);
        const size_t entries = n->children.size() - n->count("scaling");
        if (entries < rows)
            throw std::runtime_error("Missing entries in matrix type: " + name + "!");
        if (entries > rows)
            throw std::runtime_error("Redundant entries in matrix type: " + name + "!");
        for (size_t i = 0; i < rows; i++)
        {
            const InfoNode *e = n->child("(" + std::to_string(i) + ")");
            if (!e)
                throw std::runtime_error("Failed to load matrix type: " + name + "!");
            out[i] = std::strtod(e->value.c_str(), nullptr) * scaling;
        }
    }

  private:
    InfoNode root;
};

} // namespace oracle
